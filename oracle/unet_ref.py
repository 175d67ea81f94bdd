"""ORACLE (test infrastructure only - never imported by the product path).

CPU fp32 restatement of the UNet2DConditionModel forward the reference calls at
/root/reference main.py:329-336, with the peft-0.10.0 LoRA adapters the reference
injects at trainer/optimizer.py:84-95 (targets to_k,to_q,to_v,to_out.0,conv2) and the
DAAM cross-attention score side output of trainer/ti_cross_attn_loss.py:197-212.

The arithmetic of the UNet itself lives in diffusers==0.29.2 (pyproject.toml:6), which is
NOT vendored under /root/reference and not installed here.  The topology below restates
the published diffusers architecture; it is pinned by the
known-answer parameter counts 859.5 M (SD1.5) / 2567.5 M (SDXL) and the LoRA counts
128+22 / 560+17 adapted layers, 6.41 M / 25.43 M LoRA params at r=16,
checked in tests/test_oracle_unet.py.  PARITY UNPINNED at the diffusers boundary: the
reference holds no golden vector for the UNet output.

Everything is plain torch on CPU, NCHW like diffusers, weights keyed by the diffusers
state-dict names so the product loader and the kohya exporter can be checked against it.
"""
import math
from collections import OrderedDict

import torch
import torch.nn.functional as F

# ----------------------------------------------------------------------------- configs

CONFIGS = {
    "sd15": dict(
        block_out_channels=(320, 640, 1280, 1280),
        down_has_attn=(True, True, True, False),
        up_has_attn=(False, True, True, True),
        layers_per_block=2,
        transformer_layers=(1, 1, 1, 1),
        heads=(8, 8, 8, 8),              # diffusers "attention_head_dim=8" is a head COUNT
        cross_dim=768,
        linear_proj=False,
        addition=False,
        in_channels=4, out_channels=4,
        scaling_factor=0.18215,
    ),
    "sdxl": dict(
        block_out_channels=(320, 640, 1280),
        down_has_attn=(False, True, True),
        up_has_attn=(True, True, False),
        layers_per_block=2,
        transformer_layers=(1, 2, 10),
        heads=(5, 10, 20),
        cross_dim=2048,
        linear_proj=True,
        addition=True, addition_time_embed_dim=256, proj_class_in=2816,
        in_channels=4, out_channels=4,
        scaling_factor=0.13025,
    ),
    # tiny topologies with the same wiring, for fast CPU tests of host logic
    "tiny15": dict(
        block_out_channels=(64, 128, 128),
        down_has_attn=(True, True, False),
        up_has_attn=(False, True, True),
        layers_per_block=1,
        transformer_layers=(1, 1, 1),
        heads=(2, 2, 2),
        cross_dim=64,
        linear_proj=False,
        addition=False,
        in_channels=4, out_channels=4,
        scaling_factor=0.18215,
    ),
    "tinyxl": dict(
        block_out_channels=(64, 128, 128),
        down_has_attn=(False, True, True),
        up_has_attn=(True, True, False),
        layers_per_block=1,
        transformer_layers=(1, 1, 2),
        heads=(1, 2, 2),                 # head_dim 64 like SDXL
        cross_dim=128,
        linear_proj=True,
        addition=True, addition_time_embed_dim=32, proj_class_in=64 + 6 * 32,
        in_channels=4, out_channels=4,
        scaling_factor=0.13025,
    ),
}

TIME_DIM_MULT = 4  # time_embed_dim = 4 * block_out_channels[0]


def param_shapes(cfg):
    """OrderedDict name -> shape in diffusers state-dict naming."""
    P = OrderedDict()
    boc = cfg["block_out_channels"]
    c0 = boc[0]
    tdim = c0 * TIME_DIM_MULT

    def lin(n, i, o, bias=True):
        P[n + ".weight"] = (o, i)
        if bias:
            P[n + ".bias"] = (o,)

    def conv(n, i, o, k):
        P[n + ".weight"] = (o, i, k, k)
        P[n + ".bias"] = (o,)

    def norm(n, c):
        P[n + ".weight"] = (c,)
        P[n + ".bias"] = (c,)

    def resnet(n, i, o):
        norm(n + ".norm1", i)
        conv(n + ".conv1", i, o, 3)
        lin(n + ".time_emb_proj", tdim, o)
        norm(n + ".norm2", o)
        conv(n + ".conv2", o, o, 3)
        if i != o:
            conv(n + ".conv_shortcut", i, o, 1)

    def transformer(n, c, nlayers):
        norm(n + ".norm", c)
        if cfg["linear_proj"]:
            lin(n + ".proj_in", c, c)
        else:
            conv(n + ".proj_in", c, c, 1)
        for k in range(nlayers):
            b = f"{n}.transformer_blocks.{k}"
            norm(b + ".norm1", c)
            for a, kv in (("attn1", c), ("attn2", cfg["cross_dim"])):
                lin(f"{b}.{a}.to_q", c, c, bias=False)
                lin(f"{b}.{a}.to_k", kv, c, bias=False)
                lin(f"{b}.{a}.to_v", kv, c, bias=False)
                lin(f"{b}.{a}.to_out.0", c, c)
                if a == "attn1":
                    norm(b + ".norm2", c)
            norm(b + ".norm3", c)
            lin(b + ".ff.net.0.proj", c, 8 * c)
            lin(b + ".ff.net.2", 4 * c, c)
        if cfg["linear_proj"]:
            lin(n + ".proj_out", c, c)
        else:
            conv(n + ".proj_out", c, c, 1)

    conv("conv_in", cfg["in_channels"], c0, 3)
    lin("time_embedding.linear_1", c0, tdim)
    lin("time_embedding.linear_2", tdim, tdim)
    if cfg["addition"]:
        lin("add_embedding.linear_1", cfg["proj_class_in"], tdim)
        lin("add_embedding.linear_2", tdim, tdim)

    L = cfg["layers_per_block"]
    out_c = c0
    for i, c in enumerate(boc):
        in_c, out_c = out_c, c
        for j in range(L):
            resnet(f"down_blocks.{i}.resnets.{j}", in_c if j == 0 else out_c, out_c)
            if cfg["down_has_attn"][i]:
                transformer(f"down_blocks.{i}.attentions.{j}", out_c, cfg["transformer_layers"][i])
        if i != len(boc) - 1:
            conv(f"down_blocks.{i}.downsamplers.0.conv", out_c, out_c, 3)

    cm = boc[-1]
    resnet("mid_block.resnets.0", cm, cm)
    transformer("mid_block.attentions.0", cm, cfg["transformer_layers"][-1])
    resnet("mid_block.resnets.1", cm, cm)

    rev = list(reversed(boc))
    rev_layers = list(reversed(cfg["transformer_layers"]))
    out_c = rev[0]
    for i in range(len(boc)):
        prev = out_c
        out_c = rev[i]
        in_c = rev[min(i + 1, len(boc) - 1)]
        for j in range(L + 1):
            skip = in_c if j == L else out_c
            rin = prev if j == 0 else out_c
            resnet(f"up_blocks.{i}.resnets.{j}", rin + skip, out_c)
            if cfg["up_has_attn"][i]:
                transformer(f"up_blocks.{i}.attentions.{j}", out_c, rev_layers[i])
        if i != len(boc) - 1:
            conv(f"up_blocks.{i}.upsamplers.0.conv", out_c, out_c, 3)

    norm("conv_norm_out", c0)
    conv("conv_out", c0, cfg["out_channels"], 3)
    return P


def init_unet_state(cfg, seed=0, std=None, dtype=torch.float32):
    """Synthetic weights.  std=None -> N(0, 1/fan_in) so activations keep O(1)
    scale through the depth (what the parity tests use); std=0.02 reproduces SURVEY 8d's
    literal N(0,0.02^2).  Norm gamma ~ 1, beta ~ 0, both slightly perturbed so the affine
    paths are exercised; biases N(0, 0.02^2)."""
    g = torch.Generator().manual_seed(seed)
    sd = OrderedDict()
    for n, shp in param_shapes(cfg).items():
        is_norm = (".norm" in n or n.startswith("conv_norm_out")) and len(shp) == 1
        t = torch.randn(shp, generator=g, dtype=torch.float32)
        if len(shp) >= 2:
            fan_in = math.prod(shp[1:])
            t = t * (std if std is not None else 1.0 / math.sqrt(fan_in))
        else:
            t = t * 0.02
        if is_norm and n.endswith(".weight"):
            t = 1.0 + t
        sd[n] = t.to(dtype)
    return sd


LORA_TARGET_SUFFIXES = ("to_k", "to_q", "to_v", "to_out.0", "conv2")  # optimizer.py:84


def lora_targets(cfg):
    """Module paths the reference adapts (peft suffix match), in state-dict order."""
    out = []
    for n in param_shapes(cfg):
        if not n.endswith(".weight"):
            continue
        mod = n[: -len(".weight")]
        if any(mod.endswith("." + s) or mod == s for s in LORA_TARGET_SUFFIXES):
            out.append(mod)
    return out


def init_lora(cfg, rank, seed=0, b_std=0.0, dtype=torch.float32):
    """peft 0.10.0 init_lora_weights="gaussian": A ~ N(0,(1/r)^2), B = 0 (b_std>0 to exercise
    the adapter path).  Returns OrderedDict module -> (A, B); conv: A [r,Cin,3,3],
    B [Cout,r,1,1]."""
    g = torch.Generator().manual_seed(seed)
    shapes = param_shapes(cfg)
    lora = OrderedDict()
    for mod in lora_targets(cfg):
        w = shapes[mod + ".weight"]
        if len(w) == 4:
            a_shape, b_shape = (rank, w[1], w[2], w[3]), (w[0], rank, 1, 1)
        else:
            a_shape, b_shape = (rank, w[1]), (w[0], rank)
        A = torch.randn(a_shape, generator=g) * (1.0 / rank)
        B = torch.randn(b_shape, generator=g) * b_std
        lora[mod] = (A.to(dtype), B.to(dtype))
    return lora


def dora_weight_norm(w, A, B, s):
    """peft `_get_weight_norm`: per-output-channel L2 norm of W + s * (B A) (conv: B A reshaped to the kernel's shape), detached."""
    delta = (B.flatten(1) @ A.flatten(1)).reshape(w.shape)
    return (w + s * delta).flatten(1).norm(dim=1).detach()


def init_dora_magnitudes(cfg, sd, lora, lora_scale=1.0, jitter=0.0, seed=0):
    """(A, B) -> (A, B, m) with m = the weight norm at injection time (peft dora_init); jitter > 0 perturbs m multiplicatively so
    that tests exercise scale != 1."""
    g = torch.Generator().manual_seed(seed)
    out = OrderedDict()
    for k, (A, B) in lora.items():
        w = sd[k + ".weight"]
        m = dora_weight_norm(w.float(), A.float(), B.float(), lora_scale)
        if jitter:
            m = m * (1.0 + jitter * torch.randn(m.shape, generator=g))
        out[k] = (A, B, m.reshape(1, -1, 1, 1) if w.dim() == 4 else m)
    return out


# ----------------------------------------------------------------------------- forward

def timestep_embedding(t, dim):
    """diffusers Timesteps(dim, flip_sin_to_cos=True, downscale_freq_shift=0)."""
    half = dim // 2
    exponent = -math.log(10000.0) * torch.arange(half, dtype=torch.float32, device=t.device) / half
    emb = t.float()[:, None] * torch.exp(exponent)[None, :]
    return torch.cat([torch.cos(emb), torch.sin(emb)], dim=-1)


class _RoundBF16(torch.autograd.Function):
    """bf16 storage point of the HIP path, forward AND backward: the value is rounded to bf16 on the way in, the gradient on the way back
    (the kernels keep every activation and every activation gradient in bf16 between launches and accumulate in fp32 inside)."""

    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(g.dtype)


class _Ctx:
    def __init__(self, sd, lora, lora_scale, cfg, bf16_faithful=False):
        self.sd, self.lora, self.s, self.cfg = sd, lora or {}, lora_scale, cfg
        self.daam = []  # (name, scores[B,N,77])
        # bf16-faithful mode (tests only): same arithmetic, but every tensor the HIP path STORES in bf16 is rounded where it is stored - layer
        # outputs (after bias / residual / activation epilogues), the rank-r LoRA intermediate, normalised activations, softmax probabilities
        # and attention outputs, the GEGLU product - and so are their gradients.  Comparing the HIP path with this mode separates rounding
        # (what is left: summation order, a few double roundings) from logic; the fp32 mode stays the reference.
        self.q = _RoundBF16.apply if bf16_faithful else (lambda t: t)

    # DoRA entries are (A, B, magnitude) [3P-unverified: peft 0.10.0 tuners/lora/layer.py, LoraLayer._apply_dora and Conv2d._apply_dora]:
    #   weight_norm = || W + s * B A ||_2 over every axis but the output one, DETACHED ("treated as a constant", DoRA sec. 4.3)
    #   result = base(x) + (m / weight_norm - 1) * (x W^T) + (m / weight_norm) * s * B(A(x))          (the bias is not scaled)
    # magnitude: [N] for Linear, [1, N, 1, 1] for Conv2d; dora_init sets it to the norm at injection time (B = 0: ||W||).
    def linear(self, name, x):
        w, b = self.sd[name + ".weight"], self.sd.get(name + ".bias")
        y = F.linear(x, w, b)
        if name in self.lora:
            A, B, *m = self.lora[name]
            up = F.linear(self.q(self.s * F.linear(x, A)), B)
            if m:
                scale = m[0].reshape(-1) / dora_weight_norm(w, A, B, self.s)
                y = y + (scale - 1.0) * F.linear(x, w) + scale * up
            else:
                y = y + up
        return self.q(y)

    def conv(self, name, x, stride=1):
        w = self.sd[name + ".weight"]
        y = F.conv2d(x, w, self.sd.get(name + ".bias"), stride=stride, padding=w.shape[-1] // 2)
        if name in self.lora:
            A, B, *m = self.lora[name]
            up = F.conv2d(self.q(self.s * F.conv2d(x, A, None, stride=stride, padding=A.shape[-1] // 2)), B)
            if m:
                scale = (m[0].reshape(-1) / dora_weight_norm(w, A, B, self.s)).view(1, -1, 1, 1)
                y = y + (scale - 1.0) * F.conv2d(x, w, None, stride=stride, padding=w.shape[-1] // 2) + scale * up
            else:
                y = y + up
        return self.q(y)

    def gn(self, name, x, eps):
        return F.group_norm(x, 32, self.sd[name + ".weight"], self.sd[name + ".bias"], eps)

    def ln(self, name, x):
        return self.q(F.layer_norm(x, (x.shape[-1],), self.sd[name + ".weight"], self.sd[name + ".bias"], 1e-5))


def _resnet(c, n, x, temb):
    h = c.q(F.silu(c.gn(n + ".norm1", x, 1e-5)))
    h = c.conv(n + ".conv1", h)
    h = c.q(h + c.linear(n + ".time_emb_proj", F.silu(temb))[:, :, None, None])
    h = c.q(F.silu(c.gn(n + ".norm2", h, 1e-5)))
    h = c.conv(n + ".conv2", h)
    if (n + ".conv_shortcut.weight") in c.sd:
        x = c.conv(n + ".conv_shortcut", x)
    return c.q(x + h)


USE_SDPA = False


def _attention(c, n, x, ctx, heads, hooked):
    B, N, C = x.shape
    kv = x if ctx is None else ctx
    q = c.linear(n + ".to_q", x)
    k = c.linear(n + ".to_k", kv)
    v = c.linear(n + ".to_v", kv)
    d = C // heads
    qh = q.view(B, N, heads, d).transpose(1, 2)
    kh = k.view(B, -1, heads, d).transpose(1, 2)
    vh = v.view(B, -1, heads, d).transpose(1, 2)
    if USE_SDPA and not (ctx is not None and hooked):
        # what the reference runs for every attention without the DAAM hook (diffusers AttnProcessor2_0): the library's fused kernel.  Same
        # function as the explicit form below; only the library-path timing of bench.py switches it on.
        o = F.scaled_dot_product_attention(qh, kh, vh)
        return c.linear(n + ".to_out.0", o.transpose(1, 2).reshape(B, N, C))
    s = qh @ kh.transpose(-1, -2) / math.sqrt(d)
    if ctx is not None and hooked:
        # ti_cross_attn_loss.py:201-212: raw QK^T/sqrt(d) summed over heads, kept in graph
        c.daam.append((n, s.sum(dim=1)))
    o = c.q(c.q(torch.softmax(s, dim=-1)) @ vh)
    o = o.transpose(1, 2).reshape(B, N, C)
    return c.linear(n + ".to_out.0", o)


def _transformer(c, n, x, ctx, heads, nlayers, hooked):
    B, C, H, W = x.shape
    res = x
    h = c.q(c.gn(n + ".norm", x, 1e-6))
    if c.cfg["linear_proj"]:
        h = h.permute(0, 2, 3, 1).reshape(B, H * W, C)
        h = c.linear(n + ".proj_in", h)
    else:
        h = c.conv(n + ".proj_in", h)
        h = h.permute(0, 2, 3, 1).reshape(B, H * W, C)
    for k in range(nlayers):
        b = f"{n}.transformer_blocks.{k}"
        h = c.q(h + _attention(c, b + ".attn1", c.ln(b + ".norm1", h), None, heads, False))
        h = c.q(h + _attention(c, b + ".attn2", c.ln(b + ".norm2", h), ctx, heads, hooked))
        f = c.linear(b + ".ff.net.0.proj", c.ln(b + ".norm3", h))
        hid, gate = f.chunk(2, dim=-1)
        h = c.q(h + c.linear(b + ".ff.net.2", c.q(hid * F.gelu(gate))))
    if c.cfg["linear_proj"]:
        h = c.linear(n + ".proj_out", h)
        h = h.reshape(B, H, W, C).permute(0, 3, 1, 2)
    else:
        h = h.reshape(B, H, W, C).permute(0, 3, 1, 2)
        h = c.conv(n + ".proj_out", h)
    return c.q(h + res)


def unet_forward(cfg, sd, sample, timesteps, ctx, added_cond=None, lora=None, lora_scale=1.0,
                 return_daam=False, bf16_faithful=False):
    """sample [B,4,h,w], timesteps int64[B], ctx [B,77,D]; added_cond = {"text_embeds":[B,P],
    "time_ids":[B,6]} for SDXL.  Returns eps_hat [B,4,h,w] (and the list of hooked attn2 score
    maps in the reference's hook order: down_blocks then up_blocks, mid_block never hooked,
    ti_cross_attn_loss.py:97)."""
    c = _Ctx(sd, lora, lora_scale, cfg, bf16_faithful)
    if bf16_faithful:        # the engine's inputs are bf16 tensors too (noisy latent, text conditioning)
        sample, ctx = c.q(sample), c.q(ctx)
    boc = cfg["block_out_channels"]
    L = cfg["layers_per_block"]

    temb = timestep_embedding(timesteps, boc[0])
    emb = c.linear("time_embedding.linear_2", F.silu(c.linear("time_embedding.linear_1", temb)))
    if cfg["addition"]:
        tid = added_cond["time_ids"]
        te = timestep_embedding(tid.flatten(), cfg["addition_time_embed_dim"]).reshape(tid.shape[0], -1)
        add = torch.cat([added_cond["text_embeds"].float(), te], dim=-1)
        emb = emb + c.linear("add_embedding.linear_2", F.silu(c.linear("add_embedding.linear_1", add)))

    h = c.conv("conv_in", sample)
    skips = [h]
    for i in range(len(boc)):
        for j in range(L):
            h = _resnet(c, f"down_blocks.{i}.resnets.{j}", h, emb)
            if cfg["down_has_attn"][i]:
                h = _transformer(c, f"down_blocks.{i}.attentions.{j}", h, ctx, cfg["heads"][i],
                                 cfg["transformer_layers"][i], True)
            skips.append(h)
        if i != len(boc) - 1:
            h = c.conv(f"down_blocks.{i}.downsamplers.0.conv", h, stride=2)
            skips.append(h)

    h = _resnet(c, "mid_block.resnets.0", h, emb)
    h = _transformer(c, "mid_block.attentions.0", h, ctx, cfg["heads"][-1], cfg["transformer_layers"][-1], False)
    h = _resnet(c, "mid_block.resnets.1", h, emb)

    rev_heads = list(reversed(cfg["heads"]))
    rev_layers = list(reversed(cfg["transformer_layers"]))
    for i in range(len(boc)):
        for j in range(L + 1):
            h = torch.cat([h, skips.pop()], dim=1)
            h = _resnet(c, f"up_blocks.{i}.resnets.{j}", h, emb)
            if cfg["up_has_attn"][i]:
                h = _transformer(c, f"up_blocks.{i}.attentions.{j}", h, ctx, rev_heads[i], rev_layers[i], True)
        if i != len(boc) - 1:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = c.conv(f"up_blocks.{i}.upsamplers.0.conv", h)

    h = c.q(F.silu(c.gn("conv_norm_out", h, 1e-5)))
    out = c.conv("conv_out", h)
    if return_daam:
        return out, c.daam
    return out
