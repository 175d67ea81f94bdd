"""Static description of the two UNet2DConditionModel topologies the reference trains (SD1.5 / SDXL;
/root/reference trainer/models.py:15-32 loads them through diffusers `from_single_file`), in diffusers
state-dict naming, plus the analytical FLOP census the benchmark's roofline figure uses.

`tiny15` / `tinyxl` keep the exact wiring at toy sizes (channels still multiples of 64) for fast tests.
"""
import math
from collections import OrderedDict

CONFIGS = {
    "sd15": dict(block_out_channels=(320, 640, 1280, 1280), down_has_attn=(True, True, True, False),
                 up_has_attn=(False, True, True, True), layers_per_block=2, transformer_layers=(1, 1, 1, 1),
                 heads=(8, 8, 8, 8), cross_dim=768, linear_proj=False, addition=False, in_channels=4, out_channels=4,
                 scaling_factor=0.18215),
    "sdxl": dict(block_out_channels=(320, 640, 1280), down_has_attn=(False, True, True), up_has_attn=(True, True, False),
                 layers_per_block=2, transformer_layers=(1, 2, 10), heads=(5, 10, 20), cross_dim=2048, linear_proj=True,
                 addition=True, addition_time_embed_dim=256, proj_class_in=2816, in_channels=4, out_channels=4,
                 scaling_factor=0.13025),
    "tiny15": dict(block_out_channels=(64, 128, 128), down_has_attn=(True, True, False), up_has_attn=(False, True, True),
                   layers_per_block=1, transformer_layers=(1, 1, 1), heads=(2, 2, 2), cross_dim=64, linear_proj=False,
                   addition=False, in_channels=4, out_channels=4, scaling_factor=0.18215),
    "tinyxl": dict(block_out_channels=(64, 128, 128), down_has_attn=(False, True, True), up_has_attn=(True, True, False),
                   layers_per_block=1, transformer_layers=(1, 1, 2), heads=(1, 2, 2), cross_dim=128, linear_proj=True,
                   addition=True, addition_time_embed_dim=32, proj_class_in=64 + 6 * 32, in_channels=4, out_channels=4,
                   scaling_factor=0.13025),
}
TIME_DIM_MULT = 4


def diffusers_unet_config(cfg):
    """The topology as the `config.json` diffusers writes beside a saved UNet2DConditionModel (`save_pretrained`, trainer/checkpoint.py:210-212)
    - the constructor arguments that differ between SD1.5 and SDXL plus the shared ones [3P-unverified: diffusers 0.29.2 field names]."""
    boc = list(cfg["block_out_channels"])
    xl = bool(cfg["addition"])
    heads = list(cfg["heads"])
    out = {
        "_class_name": "UNet2DConditionModel", "_diffusers_version": "0.29.2", "act_fn": "silu", "in_channels": cfg["in_channels"],
        "out_channels": cfg["out_channels"], "block_out_channels": boc, "layers_per_block": cfg["layers_per_block"],
        "down_block_types": ["CrossAttnDownBlock2D" if a else "DownBlock2D" for a in cfg["down_has_attn"]],
        "up_block_types": ["CrossAttnUpBlock2D" if a else "UpBlock2D" for a in cfg["up_has_attn"]],
        "mid_block_type": "UNetMidBlock2DCrossAttn", "cross_attention_dim": cfg["cross_dim"], "norm_num_groups": 32, "norm_eps": 1e-5,
        "center_input_sample": False, "flip_sin_to_cos": True, "freq_shift": 0, "downsample_padding": 1, "sample_size": 128 if xl else 64,
        "use_linear_projection": bool(cfg["linear_proj"]), "upcast_attention": None if xl else False, "resnet_time_scale_shift": "default",
        # diffusers keeps the head COUNT under the historical name attention_head_dim (SD1.5: 8 heads everywhere, SDXL: 5 / 10 / 20)
        "attention_head_dim": heads if len(set(heads)) > 1 else heads[0],
        "transformer_layers_per_block": list(cfg["transformer_layers"]) if xl else 1,
        "addition_embed_type": "text_time" if xl else None, "addition_time_embed_dim": cfg.get("addition_time_embed_dim") if xl else None,
        "projection_class_embeddings_input_dim": cfg.get("proj_class_in") if xl else None,
    }
    return out
LORA_TARGET_SUFFIXES = ("to_k", "to_q", "to_v", "to_out.0", "conv2")  # trainer/optimizer.py:84


def _walk(cfg, on_resnet, on_transformer, on_conv, on_linear, on_norm):
    """Calls the visitors in diffusers module order; shared by param_shapes() and flops()."""
    boc = cfg["block_out_channels"]
    c0, L = boc[0], cfg["layers_per_block"]
    tdim = c0 * TIME_DIM_MULT
    on_conv("conv_in", cfg["in_channels"], c0, 3, 0)
    on_linear("time_embedding.linear_1", c0, tdim, True, None)
    on_linear("time_embedding.linear_2", tdim, tdim, True, None)
    if cfg["addition"]:
        on_linear("add_embedding.linear_1", cfg["proj_class_in"], tdim, True, None)
        on_linear("add_embedding.linear_2", tdim, tdim, True, None)
    out_c, lvl = c0, 0
    for i, c in enumerate(boc):
        in_c, out_c = out_c, c
        for j in range(L):
            on_resnet(f"down_blocks.{i}.resnets.{j}", in_c if j == 0 else out_c, out_c, lvl)
            if cfg["down_has_attn"][i]:
                on_transformer(f"down_blocks.{i}.attentions.{j}", out_c, cfg["transformer_layers"][i], cfg["heads"][i], lvl)
        if i != len(boc) - 1:
            on_conv(f"down_blocks.{i}.downsamplers.0.conv", out_c, out_c, 3, lvl + 1)
            lvl += 1
    cm = boc[-1]
    on_resnet("mid_block.resnets.0", cm, cm, lvl)
    on_transformer("mid_block.attentions.0", cm, cfg["transformer_layers"][-1], cfg["heads"][-1], lvl)
    on_resnet("mid_block.resnets.1", cm, cm, lvl)
    rev, rev_l, rev_h = list(reversed(boc)), list(reversed(cfg["transformer_layers"])), list(reversed(cfg["heads"]))
    out_c = rev[0]
    for i in range(len(boc)):
        prev, out_c = out_c, rev[i]
        in_c = rev[min(i + 1, len(boc) - 1)]
        for j in range(L + 1):
            skip = in_c if j == L else out_c
            on_resnet(f"up_blocks.{i}.resnets.{j}", (prev if j == 0 else out_c) + skip, out_c, lvl)
            if cfg["up_has_attn"][i]:
                on_transformer(f"up_blocks.{i}.attentions.{j}", out_c, rev_l[i], rev_h[i], lvl)
        if i != len(boc) - 1:
            lvl -= 1
            on_conv(f"up_blocks.{i}.upsamplers.0.conv", out_c, out_c, 3, lvl)
    on_norm("conv_norm_out", c0)
    on_conv("conv_out", c0, cfg["out_channels"], 3, 0)


def param_shapes(cfg):
    """OrderedDict diffusers-name -> shape."""
    P = OrderedDict()
    tdim = cfg["block_out_channels"][0] * TIME_DIM_MULT

    def lin(n, i, o, bias, lvl):
        P[n + ".weight"] = (o, i)
        if bias:
            P[n + ".bias"] = (o,)

    def conv(n, i, o, k, lvl):
        P[n + ".weight"] = (o, i, k, k)
        P[n + ".bias"] = (o,)

    def norm(n, c):
        P[n + ".weight"] = (c,)
        P[n + ".bias"] = (c,)

    def resnet(n, i, o, lvl):
        norm(n + ".norm1", i)
        conv(n + ".conv1", i, o, 3, lvl)
        lin(n + ".time_emb_proj", tdim, o, True, None)
        norm(n + ".norm2", o)
        conv(n + ".conv2", o, o, 3, lvl)
        if i != o:
            conv(n + ".conv_shortcut", i, o, 1, lvl)

    def transformer(n, c, nlayers, heads, lvl):
        norm(n + ".norm", c)
        (lin if cfg["linear_proj"] else (lambda a, b, d, e, f: conv(a, b, d, 1, f)))(n + ".proj_in", c, c, True, lvl)
        for k in range(nlayers):
            b = f"{n}.transformer_blocks.{k}"
            norm(b + ".norm1", c)
            for a, kvd in (("attn1", c), ("attn2", cfg["cross_dim"])):
                lin(f"{b}.{a}.to_q", c, c, False, lvl)
                lin(f"{b}.{a}.to_k", kvd, c, False, lvl)
                lin(f"{b}.{a}.to_v", kvd, c, False, lvl)
                lin(f"{b}.{a}.to_out.0", c, c, True, lvl)
                if a == "attn1":
                    norm(b + ".norm2", c)
            norm(b + ".norm3", c)
            lin(b + ".ff.net.0.proj", c, 8 * c, True, lvl)
            lin(b + ".ff.net.2", 4 * c, c, True, lvl)
        (lin if cfg["linear_proj"] else (lambda a, b, d, e, f: conv(a, b, d, 1, f)))(n + ".proj_out", c, c, True, lvl)

    _walk(cfg, resnet, transformer, conv, lin, norm)
    return P


def lora_targets(cfg):
    shapes = param_shapes(cfg)
    return [n[:-7] for n in shapes if n.endswith(".weight") and any(n[:-7].endswith("." + s) for s in LORA_TARGET_SUFFIXES)]


def fwd_flops(cfg, B, H, W, rank=16, n_ctx=77):
    """2*MAC count of every GEMM / conv / attention matmul of one UNet forward (incl. LoRA and the DAAM score
    GEMMs of the hooked cross-attentions), latent size HxW.  Reproduces SURVEY.md Appendix B:
    SDXL 128x128 B=1 r=16 -> 6.832 TFLOP."""
    tot = {"conv3": 0.0, "proj": 0.0, "attn_proj": 0.0, "attn_core": 0.0, "ff": 0.0, "lora": 0.0, "daam": 0.0}

    def px(lvl):
        return B * (H >> lvl) * (W >> lvl)

    def conv(n, i, o, k, lvl):
        key = "conv3" if k == 3 else "proj"
        tot[key] += 2.0 * px(lvl) * i * o * k * k

    def lin(n, i, o, bias, lvl):
        if lvl is None:
            tot["proj"] += 2.0 * B * i * o
        else:
            tot["proj"] += 2.0 * px(lvl) * i * o

    def norm(n, c):
        pass

    def resnet(n, i, o, lvl):
        conv(n, i, o, 3, lvl)
        conv(n, o, o, 3, lvl)
        tot["lora"] += 2.0 * px(lvl) * rank * (9 * o + o)
        tot["proj"] += 2.0 * B * (cfg["block_out_channels"][0] * TIME_DIM_MULT) * o
        if i != o:
            conv(n, i, o, 1, lvl)

    def transformer(n, c, nlayers, heads, lvl):
        N = px(lvl)
        tot["proj"] += 2 * 2.0 * N * c * c
        hooked = not n.startswith("mid_block")
        for _ in range(nlayers):
            tot["attn_proj"] += 2.0 * N * c * c * 4            # attn1 q,k,v,out
            tot["attn_proj"] += 2.0 * N * c * c * 2            # attn2 q,out
            tot["attn_proj"] += 2.0 * B * n_ctx * cfg["cross_dim"] * c * 2
            tot["lora"] += 2.0 * N * rank * 2 * c * 6 + 2.0 * B * n_ctx * rank * (cfg["cross_dim"] + c) * 2
            per_b = N // B
            tot["attn_core"] += 4.0 * B * per_b * per_b * c + 4.0 * B * per_b * n_ctx * c
            if hooked:
                tot["daam"] += 2.0 * N * n_ctx * c
            tot["ff"] += 2.0 * N * c * 8 * c + 2.0 * N * 4 * c * c

    _walk(cfg, resnet, transformer, conv, lin, norm)
    tot["total"] = sum(tot.values())
    return tot


def hbm_bytes(cfg, B, H, W, rank=16, n_ctx=77):
    """ALGORITHMIC bytes per training step (forward + backward, LoRA training: dX only) of the kernel families that are bound by HBM, not by
    MFMA - every operand touched once, bf16 activations: what `bench.py`'s roofline.hbm_kernels divides by the families' measured time.
      groupnorm  forward 2 reads + 1 write of [B,HW,C] (statistics pass + apply), backward 4 reads + 1 write
      layernorm  forward 1 read + 1 write of [N,C], backward 3 reads (x, dy, residual gradient) + 1 write
      geglu      forward reads [N,8C] writes [N,4C]; backward reads [N,8C] + [N,4C], writes [N,8C]
      adamw      7 fp32 words per adapter parameter (p, g, m, v read; p, m, v written) + the bf16 operand refresh (4 B read, 2 x 2 B written)
      lora_grad  the rows the grouped dA / dB launch contracts over tokens: dY [M,N] + T [M,r] and X [M,K] + U [M,r] per adapter
    """
    tot = {"groupnorm": 0.0, "layernorm": 0.0, "geglu": 0.0, "lora_grad": 0.0}
    Rp = 16 if rank <= 16 else (32 if rank <= 32 else 64)

    def px(lvl):
        return B * (H >> lvl) * (W >> lvl)

    def gn(m, c):
        tot["groupnorm"] += 8.0 * m * c * 2

    def resnet(n, i, o, lvl):
        gn(px(lvl), i)
        gn(px(lvl), o)
        tot["lora_grad"] += 2.0 * px(lvl) * (o + o + 2 * Rp)            # conv2 adapter: dY, X (each tap of the 3x3 window re-reads cached rows), T, U

    def transformer(n, c, nlayers, heads, lvl):
        N = px(lvl)
        gn(N, c)
        for _ in range(nlayers):
            tot["layernorm"] += 3 * 6.0 * N * c * 2
            tot["geglu"] += 32.0 * N * c * 2
            tot["lora_grad"] += 6 * 2.0 * N * (2 * c + 2 * Rp) + 2 * 2.0 * B * n_ctx * (cfg["cross_dim"] + c + 2 * Rp)

    def norm(n, c):
        gn(px(0), c)

    _walk(cfg, resnet, transformer, lambda *a: None, lambda *a: None, norm)
    n_lora = sum((rank * (9 * k if conv else k) + n * rank) for (k, n, conv) in _lora_shapes(cfg))
    tot["adamw"] = 36.0 * n_lora
    tot["n_lora_params"] = n_lora
    return tot


def _lora_shapes(cfg):
    """(K, N, is_conv) of every adapted layer (trainer/optimizer.py:84 target set)."""
    shapes = param_shapes(cfg)
    out = []
    for n in lora_targets(cfg):
        w = shapes[n + ".weight"]
        out.append((w[1], w[0], len(w) == 4 and w[-1] == 3))
    return out


def n_params(cfg):
    return sum(math.prod(s) for s in param_shapes(cfg).values())


# ---------------------------------------------------------------------------------------------- text encoders
# OpenAI CLIP ViT-L/14 text tower (SD1.5 + SDXL text_encoder) and OpenCLIP ViT-bigG/14 text tower (SDXL text_encoder_2)
CLIP_CONFIGS = {
    "clip_l": dict(vocab=49408, width=768, layers=12, heads=12, mlp=3072, act="quick_gelu", proj=None),
    "clip_g": dict(vocab=49408, width=1280, layers=32, heads=20, mlp=5120, act="gelu", proj=1280),
    "tiny_l": dict(vocab=1000, width=64, layers=2, heads=1, mlp=128, act="quick_gelu", proj=None),
    "tiny_g": dict(vocab=1000, width=64, layers=3, heads=1, mlp=128, act="gelu", proj=64),
}


def clip_param_shapes(c, n_new_tokens=0, prefix="text_model."):
    """Hugging Face CLIPTextModel(WithProjection) state-dict names -> shapes (vocab grown by the TI tokens,
    embedding_handler.py:157-223)."""
    P = OrderedDict()
    D = c["width"]
    P[prefix + "embeddings.token_embedding.weight"] = (c["vocab"] + n_new_tokens, D)
    P[prefix + "embeddings.position_embedding.weight"] = (77, D)
    for i in range(c["layers"]):
        b = f"{prefix}encoder.layers.{i}."
        for nm in ("q_proj", "k_proj", "v_proj", "out_proj"):
            P[b + f"self_attn.{nm}.weight"] = (D, D)
            P[b + f"self_attn.{nm}.bias"] = (D,)
        for ln in ("layer_norm1", "layer_norm2"):
            P[b + ln + ".weight"] = (D,)
            P[b + ln + ".bias"] = (D,)
        P[b + "mlp.fc1.weight"] = (c["mlp"], D)
        P[b + "mlp.fc1.bias"] = (c["mlp"],)
        P[b + "mlp.fc2.weight"] = (D, c["mlp"])
        P[b + "mlp.fc2.bias"] = (D,)
    P[prefix + "final_layer_norm.weight"] = (D,)
    P[prefix + "final_layer_norm.bias"] = (D,)
    if c["proj"]:
        P["text_projection.weight"] = (c["proj"], D)
    return P


def clip_fwd_flops(c, B, layers_run=None, T=77):
    """2*MAC of one text-encoder forward over T tokens (projections, MLP, causal attention counted dense)."""
    D, L = c["width"], (layers_run if layers_run is not None else c["layers"])
    per_layer = 2.0 * B * T * (4 * D * D + 2 * D * c["mlp"]) + 4.0 * B * T * T * D
    return L * per_layer


# ---------------------------------------------------------------------------------------------- VAE (AutoencoderKL)
# diffusers AutoencoderKL as SD1.5 / SDXL ship it (trainer/models.py:17-32 loads it inside the pipeline); "tiny" keeps the wiring
# at toy widths.  Used for synthetic (random-init) VAEs of the latent cache / validation render when no checkpoint is configured.
VAE_CONFIGS = {
    "sd": dict(block_out_channels=(128, 256, 512, 512), layers_per_block=2, latent_channels=4, in_channels=3),
    "tiny": dict(block_out_channels=(64, 64, 128), layers_per_block=1, latent_channels=4, in_channels=3),
}


def vae_param_shapes(c):
    """diffusers-name -> shape of every AutoencoderKL parameter (encoder, quant convs, decoder)."""
    P = OrderedDict()
    boc, L, zc, ic = c["block_out_channels"], c["layers_per_block"], c["latent_channels"], c["in_channels"]

    def conv(n, i, o, k=3):
        P[n + ".weight"], P[n + ".bias"] = (o, i, k, k), (o,)

    def vec(n, ch):
        P[n + ".weight"], P[n + ".bias"] = (ch,), (ch,)

    def resnet(n, i, o):
        vec(n + ".norm1", i), conv(n + ".conv1", i, o), vec(n + ".norm2", o), conv(n + ".conv2", o, o)
        if i != o:
            conv(n + ".conv_shortcut", i, o, 1)

    def mid(n, ch):
        resnet(n + ".resnets.0", ch, ch)
        vec(n + ".attentions.0.group_norm", ch)
        for k in ("to_q", "to_k", "to_v", "to_out.0"):
            P[f"{n}.attentions.0.{k}.weight"], P[f"{n}.attentions.0.{k}.bias"] = (ch, ch), (ch,)
        resnet(n + ".resnets.1", ch, ch)

    conv("encoder.conv_in", ic, boc[0])
    prev = boc[0]
    for i, ch in enumerate(boc):
        for j in range(L):
            resnet(f"encoder.down_blocks.{i}.resnets.{j}", prev, ch)
            prev = ch
        if i + 1 < len(boc):
            conv(f"encoder.down_blocks.{i}.downsamplers.0.conv", ch, ch)
    mid("encoder.mid_block", boc[-1])
    vec("encoder.conv_norm_out", boc[-1]), conv("encoder.conv_out", boc[-1], 2 * zc)
    conv("quant_conv", 2 * zc, 2 * zc, 1), conv("post_quant_conv", zc, zc, 1)
    rev = boc[::-1]
    conv("decoder.conv_in", zc, rev[0])
    mid("decoder.mid_block", rev[0])
    prev = rev[0]
    for i, ch in enumerate(rev):
        for j in range(L + 1):
            resnet(f"decoder.up_blocks.{i}.resnets.{j}", prev, ch)
            prev = ch
        if i + 1 < len(rev):
            conv(f"decoder.up_blocks.{i}.upsamplers.0.conv", ch, ch)
    vec("decoder.conv_norm_out", rev[-1]), conv("decoder.conv_out", rev[-1], ic)
    return P
