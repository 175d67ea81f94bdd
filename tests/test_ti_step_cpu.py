"""Host-logic test (CPU) of the full textual-inversion step: text encoders (trainable token rows) -> UNet+LoRA ->
diffusion loss + token-attention (DAAM) loss + token-std regulariser -> gradients of the TI rows and of every LoRA
tensor, against an fp32 oracle composed of Hugging Face CLIP (transformers), oracle/unet_ref.py and oracle/loss_ref.py."""
import pytest
import torch

from oracle import loss_ref as L
from oracle import unet_ref as U
from tests import emu_ops

import sd_lora_trainer_amd.clip as clip_mod
import sd_lora_trainer_amd.step as step_mod
import sd_lora_trainer_amd.unet as unet_mod
from sd_lora_trainer_amd import topology

transformers = pytest.importorskip("transformers")

V, NTOK, EOS, BOS = 203, 3, 199, 198
TRAIN_IDS = [200, 201, 202]


def _hf(act, with_proj, hidden, heads, seed, proj=64):
    from transformers import CLIPTextConfig, CLIPTextModel, CLIPTextModelWithProjection
    torch.manual_seed(seed)
    cfg = CLIPTextConfig(vocab_size=V, hidden_size=hidden, intermediate_size=2 * hidden, num_hidden_layers=3, num_attention_heads=heads,
                         max_position_embeddings=77, hidden_act=act, projection_dim=proj, eos_token_id=EOS, bos_token_id=BOS, pad_token_id=EOS)
    m = (CLIPTextModelWithProjection if with_proj else CLIPTextModel)(cfg).eval()
    for p in m.parameters():
        if p.dim() == 1:
            p.data.add_(0.05 * torch.randn_like(p))
    return m


def _captions(B):
    lists = [[BOS, 5, 17, 33] + TRAIN_IDS + [41, EOS], [BOS, 7, 9, EOS]][:B]     # 2nd caption lacks the TI tokens
    ids = torch.full((B, 77), EOS, dtype=torch.int64)
    for b, l in enumerate(lists):
        ids[b, :len(l)] = torch.tensor(l)
    return lists, ids


@pytest.mark.parametrize("version,B", [("tiny15", 2), ("tinyxl", 1), ("tinyxl", 2)])
def test_ti_step_matches_oracle(version, B):
    cfg = U.CONFIGS[version]
    xl = cfg["addition"]
    rank, h, w_ta, w_std = 4, 16, 2e-2, 0.01
    sd = U.init_unet_state(cfg, seed=0)
    lora = U.init_lora(cfg, rank, seed=1, b_std=0.05)
    if xl:
        hf = [_hf("quick_gelu", False, 64, 1, 11), _hf("gelu", True, 64, 1, 12, proj=cfg["proj_class_in"] - 6 * cfg["addition_time_embed_dim"])]
    else:
        hf = [_hf("quick_gelu", False, 64, 2, 11)]
    g = torch.Generator().manual_seed(3)
    latent = torch.randn(B, 4, h, h, generator=g) * cfg["scaling_factor"]
    noise = torch.randn(B, 4, h, h, generator=g)
    mask = (torch.rand(B, 1, h, h, generator=g) * 0.95 + 0.05).repeat(1, 4, 1, 1).contiguous()
    t = torch.tensor([10, 900][:B])
    tid = torch.tensor([[1024., 1024, 0, 0, 128, 128]] * B) if xl else None
    lists, ids = _captions(B)

    # ------------------------------------------------------------------ oracle (autograd)
    embs = [m.get_input_embeddings().weight for m in hf]
    outs = [m(input_ids=ids, output_hidden_states=True) for m in hf]
    if xl:
        ctx = torch.cat([outs[0].hidden_states[-2], outs[1].hidden_states[-2]], dim=-1)
        add = {"text_embeds": outs[1].text_embeds, "time_ids": tid}
    else:
        ctx, add = outs[0].last_hidden_state, None
    lora_g, params = {}, []
    for k, (A, Bm) in lora.items():
        A, Bm = A.clone().requires_grad_(True), Bm.clone().requires_grad_(True)
        lora_g[k] = (A, Bm)
        params += [A, Bm]
    acp = L.ddpm_alphas_cumprod()
    noisy = L.add_noise(acp, latent, noise, t)
    pred, daam = U.unet_forward(cfg, sd, noisy, t, ctx, add, lora=lora_g, return_daam=True)
    img_loss = L.diffusion_loss(pred, noise, noisy, mask, acp, t, snr_gamma=5.0)
    ta = L.token_attention_loss(L.daam_stack([s for _, s in daam], 1.0), mask, lists, TRAIN_IDS)
    reg = torch.stack([L.DistributionStats(e.detach()[:-NTOK]).std_loss(e[-NTOK:]) for e in embs]).mean()
    loss = img_loss + w_ta * ta + w_std * reg
    grads = torch.autograd.grad(loss, params + embs)
    g_lora = {k: (grads[2 * i], grads[2 * i + 1]) for i, k in enumerate(lora)}
    g_rows = [ge[-NTOK:] for ge in grads[len(params):]]

    # ------------------------------------------------------------------ the plan, through the op emulation
    rt = unet_mod.Runtime("cpu", B, act_dtype=torch.float32, ops=emu_ops)
    unet = unet_mod.UNet(rt, topology.CONFIGS[version], sd, lora_rank=rank)
    unet.arena.load(lora)
    sds = [{k: v.detach() for k, v in m.state_dict().items()} for m in hf]
    if xl:
        encs = [clip_mod.ClipTextEncoder(rt, "te1", sds[0], heads=1, act="quick_gelu", mode="penultimate", with_projection=False, n_train=NTOK),
                clip_mod.ClipTextEncoder(rt, "te2", sds[1], heads=1, act="gelu", mode="penultimate", with_projection=True, n_train=NTOK)]
    else:
        encs = [clip_mod.ClipTextEncoder(rt, "te1", sds[0], heads=2, act="quick_gelu", mode="last", with_projection=False, n_train=NTOK)]
    text = step_mod.TextStack(rt, encs, pool_mode="first_eos", eos_token_id=EOS)
    ts = step_mod.TrainStep(rt, unet, latent_hw=(h, h), snr_gamma=5.0, l1_penalty=0.0, weight_decay=0.0, text=text, n_tokens=NTOK,
                            token_attention_loss_w=w_ta, ti_std_loss_w=w_std)
    ts.set_batch(latent, noise, t, mask, time_ids=tid, ids=[ids] * len(encs), caption_token_lists=lists)
    ts.forward_backward()
    torch.testing.assert_close(ts.loss[0], img_loss.detach(), rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(ts.ta.loss[0], ta.detach(), rtol=1e-3, atol=1e-5)
    torch.testing.assert_close(ts.ti.reg_loss[0], (w_std * reg).detach(), rtol=1e-3, atol=1e-7)
    for got, ref in zip(ts.ti.grad_rows, g_rows):
        scale = float(ref.abs().max())
        assert scale > 0 and float((got - ref).abs().max()) <= 3e-3 * scale, (float((got - ref).abs().max()), scale)
    got = unet.arena.export("grads")
    for k in g_lora:
        for a, b_ in zip(got[k], g_lora[k]):
            scale = max(float(b_.abs().max()), 1e-8)
            assert float((a - b_).abs().max()) <= 3e-3 * scale + 1e-7, (k, float((a - b_).abs().max()), scale)

    # one TI optimiser step: rows-only AdamW == torch.optim.AdamW on the rows
    rows0 = [r.clone() for r in ts.ti.rows]
    grows = [r.clone() for r in ts.ti.grad_rows]
    ts.set_hyper(1e-3, lr_ti=1e-3)
    ts.optimizer_step()
    for r0, gr, r1, enc in zip(rows0, grows, ts.ti.rows, encs):
        p, m, v = r0.clone(), torch.zeros_like(r0), torch.zeros_like(r0)
        L.adamw_step(p, gr, m, v, 1, 1e-3, weight_decay=0.0)
        torch.testing.assert_close(r1, p, rtol=1e-5, atol=1e-7)
        torch.testing.assert_close(enc.table[-NTOK:].float(), p, rtol=1e-5, atol=1e-7)    # gathered table was refreshed


@pytest.mark.parametrize("version,B,rank,dora,w_tok", [("tiny15", 2, 4, False, 0.0), ("tinyxl", 1, 16, False, 0.0), ("tiny15", 1, 24, False, 0.0),
                                                       ("tinyxl", 1, 16, True, 0.0), ("tiny15", 2, 24, True, 0.0),
                                                       ("tinyxl", 1, 16, False, 2e-3), ("tiny15", 2, 24, False, 2e-3),     # + tok_cond_reg_w: a second pass through the adapters
                                                       ("tinyxl", 1, 16, True, 2e-3), ("tiny15", 2, 24, True, 2e-3)])      # ... through weight-decomposed ones (round 6)
def test_text_encoder_lora_matches_oracle(version, B, rank, dora, w_tok):
    """a21 (`text_encoder_lora_optimizer`, trainer/optimizer.py:157-202): peft LoRA on q/k/v/out_proj of every text-encoder
    layer, trained by its own AdamW next to TI and the UNet LoRA.  Oracle: Hugging Face CLIP called functionally with
    W + (alpha/r) B A in place of the four projection weights, autograd for dA / dB."""
    from torch.func import functional_call
    import sd_lora_trainer_amd.checkpoint as ckpt
    cfg = U.CONFIGS[version]
    xl = cfg["addition"]
    h, w_ta, w_std = 16, 2e-2, 0.01
    sd = U.init_unet_state(cfg, seed=0)
    lora = U.init_lora(cfg, 4, seed=1, b_std=0.05)
    if xl:
        hf = [_hf("quick_gelu", False, 64, 1, 11), _hf("gelu", True, 64, 1, 12, proj=cfg["proj_class_in"] - 6 * cfg["addition_time_embed_dim"])]
    else:
        hf = [_hf("quick_gelu", False, 64, 2, 11)]
    g = torch.Generator().manual_seed(3)
    latent = torch.randn(B, 4, h, h, generator=g) * cfg["scaling_factor"]
    noise = torch.randn(B, 4, h, h, generator=g)
    mask = (torch.rand(B, 1, h, h, generator=g) * 0.95 + 0.05).repeat(1, 4, 1, 1).contiguous()
    t = torch.tensor([10, 900][:B])
    tid = torch.tensor([[1024., 1024, 0, 0, 128, 128]] * B) if xl else None
    lists, ids = _captions(B)

    # ------------------------------------------------------------------ the plan (built first: it names the adapters)
    rt = unet_mod.Runtime("cpu", B, act_dtype=torch.float32, ops=emu_ops)
    unet = unet_mod.UNet(rt, topology.CONFIGS[version], sd, lora_rank=4)
    unet.arena.load(lora)
    te_arena = unet_mod.LoraArena(rt, rank, 1.0, problems=[], dora=dora)      # dora: use_dora reaches the text-encoder adapters too (optimizer.py:157-165)
    sds = [{k: v.detach() for k, v in m.state_dict().items()} for m in hf]
    prefixes = ["text_encoder.", "text_encoder_2."]
    kw = [dict(heads=1, act="quick_gelu", mode="penultimate", with_projection=False), dict(heads=1, act="gelu", mode="penultimate", with_projection=True)] \
        if xl else [dict(heads=2, act="quick_gelu", mode="last", with_projection=False)]
    encs = [clip_mod.ClipTextEncoder(rt, f"te{i + 1}", sds[i], n_train=NTOK, arena=te_arena, lora_prefix=prefixes[i], **k) for i, k in enumerate(kw)]
    te_arena.finalize()
    gl = torch.Generator().manual_seed(21)
    te_lora = {e["name"]: (torch.randn(rank, e["K"], generator=gl) / rank, torch.randn(e["N"], rank, generator=gl) * 0.05) for e in te_arena.entries}
    if dora:       # magnitudes: the weight norm of the merged matrix, perturbed so that the column factor is not 1
        for i_, pre in enumerate(prefixes[:len(hf)]):
            for name, (A, Bm, *_) in list(te_lora.items()):
                if name.startswith(pre) and not (i_ == 0 and name.startswith(prefixes[1])):
                    w = sds[i_][name[len(pre):] + ".weight"]
                    te_lora[name] = (A, Bm, (w + te_arena.scale * Bm @ A).norm(dim=1) * (1.0 + 0.05 * torch.randn(w.shape[0], generator=gl)))
    te_arena.load(te_lora)
    n_layers_run = [3 if (not xl or i == 1) else 2 for i in range(len(hf))]            # SDXL CLIP-L: the last layer feeds nothing
    assert len(te_arena.entries) == 4 * sum(n_layers_run)
    text = step_mod.TextStack(rt, encs, pool_mode="first_eos", eos_token_id=EOS, arena=te_arena)
    tok = list(TRAIN_IDS)
    caps = [[5, 6, 7] + tok, tok, [5, 6, 7] + tok + [8, 9] + tok, tok + [10] + tok]
    reg_ids = torch.full((4, 77), EOS, dtype=torch.int64)
    for r_, c_ in enumerate(caps):
        reg_ids[r_, 0] = BOS
        reg_ids[r_, 1:1 + len(c_)] = torch.tensor(c_)
    ts = step_mod.TrainStep(rt, unet, latent_hw=(h, h), snr_gamma=5.0, l1_penalty=0.0, weight_decay=0.0, text=text, n_tokens=NTOK,
                            token_attention_loss_w=w_ta, ti_std_loss_w=w_std, text_lora_weight_decay=1e-5,
                            tok_cond_reg_w=w_tok, reg_caption_ids=[reg_ids] * len(encs) if w_tok else None)
    ts.set_batch(latent, noise, t, mask, time_ids=tid, ids=[ids] * len(encs), caption_token_lists=lists)
    ts.forward_backward()

    # ------------------------------------------------------------------ oracle (autograd through merged projections)
    te_g, te_params, outs, routs = {}, [], [], []
    for i, m in enumerate(hf):
        over = {}
        for name, (A, Bm, *mag) in te_lora.items():
            if not name.startswith(prefixes[i]) or (i == 0 and name.startswith(prefixes[1])):
                continue
            A, Bm = A.clone().requires_grad_(True), Bm.clone().requires_grad_(True)
            key = name[len(prefixes[i]):] + ".weight"
            merged = sds[i][key] + te_arena.scale * Bm @ A
            if dora:      # peft _apply_dora, merged form: rows scaled by m / ||W + s B A|| (norm detached); the bias is not scaled
                mg = mag[0].clone().requires_grad_(True)
                te_g[name] = (A, Bm, mg)
                te_params += [A, Bm, mg]
                merged = (mg / merged.norm(dim=1).detach())[:, None] * merged
            else:
                te_g[name] = (A, Bm)
                te_params += [A, Bm]
            over[key] = merged
        outs.append(functional_call(m, over, kwargs=dict(input_ids=ids, output_hidden_states=True)))
        if w_tok:
            routs.append(functional_call(m, over, kwargs=dict(input_ids=reg_ids, output_hidden_states=True)))
    embs = [m.get_input_embeddings().weight for m in hf]
    if xl:
        ctx = torch.cat([outs[0].hidden_states[-2], outs[1].hidden_states[-2]], dim=-1)
        add = {"text_embeds": outs[1].text_embeds, "time_ids": tid}
    else:
        ctx, add = outs[0].last_hidden_state, None
    acp = L.ddpm_alphas_cumprod()
    noisy = L.add_noise(acp, latent, noise, t)
    pred, daam = U.unet_forward(cfg, sd, noisy, t, ctx, add, lora={k: v for k, v in lora.items()}, return_daam=True)
    img_loss = L.diffusion_loss(pred, noise, noisy, mask, acp, t, snr_gamma=5.0)
    ta = L.token_attention_loss(L.daam_stack([s for _, s in daam], 1.0), mask, lists, TRAIN_IDS)
    reg = torch.stack([L.DistributionStats(e.detach()[:-NTOK]).std_loss(e[-NTOK:]) for e in embs]).mean()
    total = img_loss + w_ta * ta + w_std * reg
    if w_tok:      # loss.py:207-211, 241-251: the prompt-norm target on the four trigger captions, through the adapted encoders
        rctx = torch.cat([routs[0].hidden_states[-2], routs[1].hidden_states[-2]], dim=-1) if xl else routs[0].last_hidden_state
        tokreg, tok_norm = L.prompt_norm_loss(rctx, 34.5 if xl else 27.8)
        total = total + w_tok * tokreg
        torch.testing.assert_close(ts.tok_reg_norm[0], tok_norm.detach(), rtol=1e-4, atol=0)
    grads = torch.autograd.grad(total, te_params + embs)
    torch.testing.assert_close(ts.loss[0], img_loss.detach(), rtol=1e-4, atol=1e-6)
    got = te_arena.export("grads")
    assert set(got) == set(te_g)
    gmax = max(float(x.abs().max()) for x in grads[:len(te_params)])
    assert gmax > 0
    npt = 3 if dora else 2
    for i, name in enumerate(te_g):
        for a, b_ in zip(got[name], grads[npt * i: npt * i + npt]):
            assert float((a - b_).abs().max()) <= 3e-3 * float(b_.abs().max()) + 1e-6 * gmax, (name, float((a - b_).abs().max()), float(b_.abs().max()))
    for got_r, ref in zip(ts.ti.grad_rows, [ge[-NTOK:] for ge in grads[len(te_params):]]):
        assert float((got_r - ref).abs().max()) <= 3e-3 * float(ref.abs().max())

    # one optimiser step: the text-encoder arena follows AdamW with its own lr / weight decay, the other groups theirs
    p0, g0 = te_arena.params.clone(), te_arena.grads.clone()
    ts.set_hyper(1e-3, lr_ti=1e-3, lr_te=2e-4)
    ts.optimizer_step()
    pref, m_, v_ = p0.clone(), torch.zeros_like(p0), torch.zeros_like(p0)
    L.adamw_step(pref, g0, m_, v_, 1, 2e-4, weight_decay=1e-5)
    torch.testing.assert_close(te_arena.params, pref, rtol=1e-5, atol=1e-8)
    e0 = te_arena.entries[0]
    torch.testing.assert_close(e0["A_s"][:rank].float(), e0["A"], rtol=1e-6, atol=0)          # compute copies refreshed
    # kohya keys of the text-encoder adapters in the checkpoint file
    ksd = ckpt.lora_to_kohya(te_arena.export(), key=ckpt.kohya_text_key)
    k0 = ckpt.kohya_text_key(e0["name"])
    assert k0.startswith("lora_te1_") and k0.endswith("encoder_layers_0_self_attn_q_proj") and k0 + ".lora_down.weight" in ksd
    if xl:
        assert any(k.startswith("lora_te2_") for k in ksd)


@pytest.mark.parametrize("version", ["tiny15", "tinyxl"])
def test_token_warmup_matches_oracle(version):
    """a20 `pre_optimize_token_embeddings` (trainer/embedding_handler.py:321-399): k AdamW steps on the token rows against
    the same loop written with Hugging Face CLIP + autograd + torch.optim.AdamW on the full (gradient-masked) tables."""
    cfg = U.CONFIGS[version]
    xl = cfg["addition"]
    B, h, steps, lr = 2, 16, 4, 2e-3
    if xl:
        hf = [_hf("quick_gelu", False, 64, 1, 11), _hf("gelu", True, 64, 1, 12, proj=cfg["proj_class_in"] - 6 * cfg["addition_time_embed_dim"])]
    else:
        hf = [_hf("quick_gelu", False, 64, 2, 11)]
    prompt = torch.full((77,), EOS, dtype=torch.int64)
    prompt[:5] = torch.tensor([BOS] + TRAIN_IDS + [EOS])
    target = torch.full((77,), EOS, dtype=torch.int64)
    target[:6] = torch.tensor([BOS, 5, 17, 33, 41, EOS])

    rt = unet_mod.Runtime("cpu", B, act_dtype=torch.float32, ops=emu_ops)
    unet = unet_mod.UNet(rt, topology.CONFIGS[version], U.init_unet_state(cfg, seed=0), lora_rank=4)
    sds = [{k: v.detach().clone() for k, v in m.state_dict().items()} for m in hf]
    kw = [dict(heads=1, act="quick_gelu", mode="penultimate", with_projection=False), dict(heads=1, act="gelu", mode="penultimate", with_projection=True)] \
        if xl else [dict(heads=2, act="quick_gelu", mode="last", with_projection=False)]
    encs = [clip_mod.ClipTextEncoder(rt, f"te{i + 1}", sds[i], n_train=NTOK, **k) for i, k in enumerate(kw)]
    text = step_mod.TextStack(rt, encs, pool_mode="first_eos", eos_token_id=EOS)
    ts = step_mod.TrainStep(rt, unet, latent_hw=(h, h), text=text, n_tokens=NTOK)
    got_losses = ts.token_warmup([prompt] * len(encs), [target] * len(encs), steps, lr)

    embs = [m.get_input_embeddings().weight for m in hf]
    stats = [L.DistributionStats(e.detach()[:-NTOK].clone()) for e in embs]
    opt = torch.optim.AdamW(embs, lr=lr, weight_decay=0.0)

    def encode(ids):
        outs = [m(input_ids=ids.view(1, 77), output_hidden_states=True) for m in hf]
        if xl:
            return torch.cat([outs[0].hidden_states[-2], outs[1].hidden_states[-2]], dim=-1), outs[1].text_embeds
        return outs[0].last_hidden_state, None
    with torch.no_grad():
        tgt, tgt_pooled = encode(target)
    ref_losses = []
    for _ in range(steps):
        pe, pooled = encode(prompt)
        loss = 0.2 * L.target_prompt_loss(pe, tgt, pooled, tgt_pooled)
        loss = loss + 0.5 * torch.stack([st.std_loss(e[-NTOK:]) for st, e in zip(stats, embs)]).mean()
        opt.zero_grad()
        loss.backward()
        for e in embs:
            e.grad.data[:-NTOK] *= 0.0
        opt.step()
        ref_losses.append(float(loss))
    torch.testing.assert_close(torch.tensor(got_losses), torch.tensor(ref_losses), rtol=2e-4, atol=1e-6)
    for r, e, enc in zip(ts.ti.rows, embs, encs):
        moved = float((e[-NTOK:] - enc.table[-NTOK:]).abs().max())
        torch.testing.assert_close(r, e.detach()[-NTOK:], rtol=0, atol=2e-2 * steps * lr)
        assert float((r - e.detach()[-NTOK:]).abs().max()) <= 2e-2 * steps * lr, moved
    assert float(ts.ti.m.abs().max()) == 0.0                 # the warm-up optimizer's moments are not carried into training


@pytest.mark.parametrize("version", ["tiny15", "tinyxl"])
def test_optional_regularisers_match_oracle(version):
    run_optional_regularisers(version, "cpu", emu_ops, torch.float32, rel_val=1e-3, rel_grad=3e-3)


def run_optional_regularisers(version, device, ops, act_dtype, rel_val, rel_grad):
    """cond_reg_w (prompt-embedding norm, loss.py:201-205, 235-239), tok_cond_reg_w (the same on four captions around the trigger,
    loss.py:207-211, 241-251) and tok_cov_reg_w (covariance of the token rows, loss.py:213-221, 275-289), all 0 by default:
    token-row gradients of the whole step with the three terms switched on.  (Also the body of the GPU test.)"""
    cfg = U.CONFIGS[version]
    xl = cfg["addition"]
    B, rank, h, w_cond, w_cov, w_tok = 2, 4, 16, 3e-3, 50.0, 2e-3
    sd = U.init_unet_state(cfg, seed=0)
    lora = U.init_lora(cfg, rank, seed=1, b_std=0.05)
    hf = [_hf("quick_gelu", False, 64, 1, 11), _hf("gelu", True, 64, 1, 12, proj=cfg["proj_class_in"] - 6 * cfg["addition_time_embed_dim"])] if xl \
        else [_hf("quick_gelu", False, 64, 2, 11)]
    g = torch.Generator().manual_seed(3)
    latent = torch.randn(B, 4, h, h, generator=g) * cfg["scaling_factor"]
    noise = torch.randn(B, 4, h, h, generator=g)
    mask = torch.ones(B, 4, h, h)
    t = torch.tensor([10, 900])
    tid = torch.tensor([[1024., 1024, 0, 0, 128, 128]] * B) if xl else None
    lists, ids = _captions(B)
    target = 34.5 if xl else 27.8

    embs = [m.get_input_embeddings().weight for m in hf]
    outs = [m(input_ids=ids, output_hidden_states=True) for m in hf]
    if xl:
        ctx = torch.cat([outs[0].hidden_states[-2], outs[1].hidden_states[-2]], dim=-1)
        add = {"text_embeds": outs[1].text_embeds, "time_ids": tid}
    else:
        ctx, add = outs[0].last_hidden_state, None
    acp = L.ddpm_alphas_cumprod()
    noisy = L.add_noise(acp, latent, noise, t)
    pred = U.unet_forward(cfg, sd, noisy, t, ctx, add, lora={k: v for k, v in lora.items()})
    img_loss = L.diffusion_loss(pred, noise, noisy, mask, acp, t, snr_gamma=5.0)
    cond, norm_val = L.prompt_norm_loss(ctx, target)
    cov = torch.stack([L.DistributionStats(e.detach()[:-NTOK]).cov_loss(e[-NTOK:]) for e in embs]).mean()
    # tok_cond_reg_w (loss.py:207-211, 241-251): four captions around the trigger tokens, encoded with autograd
    tok = list(TRAIN_IDS)
    caps = [[5, 6, 7] + tok, tok, [5, 6, 7] + tok + [8, 9] + tok, tok + [10] + tok]
    reg_ids = torch.full((4, 77), EOS, dtype=torch.int64)
    for r, c in enumerate(caps):
        reg_ids[r, 0] = BOS
        reg_ids[r, 1:1 + len(c)] = torch.tensor(c)
    routs = [m(input_ids=reg_ids, output_hidden_states=True) for m in hf]
    rctx = torch.cat([routs[0].hidden_states[-2], routs[1].hidden_states[-2]], dim=-1) if xl else routs[0].last_hidden_state
    tokreg, tok_norm = L.prompt_norm_loss(rctx, target)
    grads = torch.autograd.grad(img_loss + w_cond * cond + w_cov * cov + w_tok * tokreg, embs)

    rt = unet_mod.Runtime(device, B, act_dtype=act_dtype, ops=ops)
    unet = unet_mod.UNet(rt, topology.CONFIGS[version], sd, lora_rank=rank)
    unet.arena.load(lora)
    sds = [{k: v.detach() for k, v in m.state_dict().items()} for m in hf]
    kw = [dict(heads=1, act="quick_gelu", mode="penultimate", with_projection=False), dict(heads=1, act="gelu", mode="penultimate", with_projection=True)] \
        if xl else [dict(heads=2, act="quick_gelu", mode="last", with_projection=False)]
    encs = [clip_mod.ClipTextEncoder(rt, f"te{i + 1}", sds[i], n_train=NTOK, **k) for i, k in enumerate(kw)]
    text = step_mod.TextStack(rt, encs, pool_mode="first_eos", eos_token_id=EOS)
    ts = step_mod.TrainStep(rt, unet, latent_hw=(h, h), snr_gamma=5.0, l1_penalty=0.0, weight_decay=0.0, text=text, n_tokens=NTOK,
                            token_attention_loss_w=0.0, ti_std_loss_w=0.0, cond_reg_w=w_cond, tok_cov_reg_w=w_cov,
                            tok_cond_reg_w=w_tok, reg_caption_ids=[reg_ids] * len(encs))
    dv = lambda x: x.to(device) if x is not None else None  # noqa: E731
    ts.set_batch(dv(latent), dv(noise), dv(t), dv(mask), time_ids=dv(tid), ids=[dv(ids)] * len(encs), caption_token_lists=lists)
    ts.forward_backward()
    rv = rel_val
    torch.testing.assert_close(ts.tok_reg_norm[0].cpu(), tok_norm.detach(), rtol=rv, atol=0)
    torch.testing.assert_close(ts.tok_reg_loss[0].cpu(), (w_tok * tokreg).detach(), rtol=3 * rv, atol=1e-8)
    torch.testing.assert_close(ts.cond_norm[0].cpu(), norm_val.detach(), rtol=rv, atol=0)
    torch.testing.assert_close(ts.cond_reg_loss[0].cpu(), (w_cond * cond).detach(), rtol=3 * rv, atol=1e-8)
    torch.testing.assert_close(ts.ti.cov_loss[0].cpu(), (w_cov * cov).detach(), rtol=3 * rv, atol=1e-9)
    for got, ref in zip(ts.ti.grad_rows, [ge[-NTOK:] for ge in grads]):
        scale = float(ref.abs().max())
        err = float((got.cpu() - ref).abs().max())
        assert scale > 0 and err <= rel_grad * scale, (err, scale)
