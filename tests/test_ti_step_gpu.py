"""End-to-end textual-inversion step on the MI355X (bf16 HIP path) against the fp32 oracle (transformers CLIP +
oracle/unet_ref.py + oracle/loss_ref.py): TI-row gradients, LoRA gradients, losses; then hipGraph replays train.
Tolerances as tests/test_step_gpu.py (bf16 noise floor): cosine >= 0.99, relative L2 <= 8e-2, losses 2e-2."""
import math

import pytest
import torch

from tests.test_ti_step_cpu import EOS, NTOK, TRAIN_IDS, _captions, _hf

pytestmark = pytest.mark.gpu


def _cos_rel(a, b):
    a, b = a.reshape(-1).double().cpu(), b.reshape(-1).double().cpu()
    return float(a @ b / (a.norm() * b.norm() + 1e-30)), float((a - b).norm() / (b.norm() + 1e-30))


@pytest.mark.parametrize("version,B,concurrent", [("tiny15", 2, False), ("tinyxl", 2, False), ("tinyxl", 2, True)])
def test_ti_step_gpu_matches_oracle(version, B, concurrent):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from oracle import loss_ref as L
    from oracle import unet_ref as U
    import sd_lora_trainer_amd.clip as clip_mod
    import sd_lora_trainer_amd.step as step_mod
    import sd_lora_trainer_amd.unet as unet_mod
    from sd_lora_trainer_amd import topology
    cfg = U.CONFIGS[version]
    xl = cfg["addition"]
    rank, w_ta, w_std = 4, 2e-2, 0.01
    h = 32 if xl else 16      # every hooked map needs a multiple of 64 tokens (K of the score-gradient GEMMs)
    sd = U.init_unet_state(cfg, seed=0)
    lora = U.init_lora(cfg, rank, seed=1, b_std=0.05)
    hf = ([_hf("quick_gelu", False, 64, 1, 11), _hf("gelu", True, 64, 1, 12, proj=cfg["proj_class_in"] - 6 * cfg["addition_time_embed_dim"])]
          if xl else [_hf("quick_gelu", False, 64, 2, 11)])
    g = torch.Generator().manual_seed(3)
    latent = torch.randn(B, 4, h, h, generator=g) * cfg["scaling_factor"]
    noise = torch.randn(B, 4, h, h, generator=g)
    mask = (torch.rand(B, 1, h, h, generator=g) * 0.95 + 0.05).repeat(1, 4, 1, 1).contiguous()
    t = torch.tensor([10, 900][:B])
    tid = torch.tensor([[1024., 1024, 0, 0, 8. * h, 8. * h]] * B) if xl else None
    lists, ids = _captions(B)
    # the HIP path stores tables/weights in bf16: round the oracle's CLIP weights the same way so both see the same model
    for m in hf:
        for p in m.parameters():
            p.data = p.data.to(torch.bfloat16).float()
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
    grads = torch.autograd.grad(img_loss + w_ta * ta + w_std * reg, params + embs)
    g_lora = torch.cat([x.reshape(-1) for x in grads[:len(params)]])
    g_rows = [ge[-NTOK:] for ge in grads[len(params):]]

    rt = unet_mod.Runtime("cuda:0", B)
    unet = unet_mod.UNet(rt, topology.CONFIGS[version], sd, lora_rank=rank)
    unet.arena.load(lora)
    sds = [{k: v.detach() for k, v in m.state_dict().items()} for m in hf]
    if xl:
        encs = [clip_mod.ClipTextEncoder(rt, "te1", sds[0], heads=1, act="quick_gelu", mode="penultimate", with_projection=False, n_train=NTOK),
                clip_mod.ClipTextEncoder(rt, "te2", sds[1], heads=1, act="gelu", mode="penultimate", with_projection=True, n_train=NTOK)]
    else:
        encs = [clip_mod.ClipTextEncoder(rt, "te1", sds[0], heads=2, act="quick_gelu", mode="last", with_projection=False, n_train=NTOK)]
    text = step_mod.TextStack(rt, encs, pool_mode="first_eos", eos_token_id=EOS, concurrent=concurrent)   # forked encoder streams + per-phase graphs
    ts = step_mod.TrainStep(rt, unet, latent_hw=(h, h), snr_gamma=5.0, l1_penalty=0.0, weight_decay=0.0, text=text, n_tokens=NTOK,
                            token_attention_loss_w=w_ta, ti_std_loss_w=w_std)
    ts.set_batch(latent.cuda(), noise.cuda(), t.cuda(), mask.cuda(), time_ids=tid.cuda() if xl else None, ids=[ids] * len(encs),
                 caption_token_lists=lists)
    ts.forward_backward()
    torch.cuda.synchronize()
    assert abs(float(ts.loss) - float(img_loss)) <= 2e-2 * float(img_loss)
    assert abs(float(ts.ta.loss) - float(ta)) <= 3e-2 * abs(float(ta))
    assert abs(float(ts.ti.reg_loss) - w_std * float(reg)) <= 2e-2 * w_std * float(reg)
    got_lora = torch.cat([x.reshape(-1) for k in unet.arena.export("grads").values() for x in k])
    cos, rel = _cos_rel(got_lora, g_lora)
    assert cos >= 0.99 and rel <= 8e-2, f"LoRA grads cos {cos} rel {rel}"
    for got, ref in zip(ts.ti.grad_rows, g_rows):
        cos, rel = _cos_rel(got, ref)
        assert cos >= 0.985 and rel <= 0.12, f"TI row grads cos {cos} rel {rel}"
    # graph replays: the loss of a fixed batch goes down when both LoRA and the token rows are trained
    ts.capture(warmup=1)
    losses = []
    for i in range(6):
        ts.run(1e-3, lr_ti=1e-3)
        losses.append(ts.total_loss())
    assert all(torch.isfinite(torch.tensor(losses))) and losses[-1] < losses[0], losses
    # frozen token embeddings (ti lr == 0, main.py:273-274): the fast path skips the text-encoder backward and the rows-only
    # AdamW - the token rows must stay bit-identical while the LoRA keeps training
    rows0 = [p.clone() for p in [ts.ti.params]]
    lora0 = unet.arena.params.clone()
    for i in range(2):
        ts.run(1e-3, lr_ti=0.0)
    torch.cuda.synchronize()
    assert all(torch.equal(a, b) for a, b in zip(rows0, [ts.ti.params]))
    assert not torch.equal(lora0, unet.arena.params)
    assert ts.grad_norm() > 0.0 and math.isfinite(ts.total_loss())
    # cached conditioning (f4): with frozen rows the conditioning of a caption is a constant - the same frozen step with the
    # conditioning handed over (no text-encoder forward in the replayed graph) gives the same loss and the same update
    if not concurrent:
        a = unet.arena
        with torch.no_grad():           # what train() does once when the rows freeze: encode the captions with the final rows
            pooled_c = ts.text.forward(ts.ctx)
            pooled_c = pooled_c.clone() if pooled_c is not None else None
        ctx_c = ts.ctx.view(B, unet_mod.CTX_PAD, -1)[:, :77].clone()
        snap, step0 = [t_.clone() for t_ in (a.params, a.m, a.v)], ts.opt_step
        ts.run(1e-3, lr_ti=0.0)
        torch.cuda.synchronize()
        l_ref, p_ref = float(ts.loss), a.params.clone()
        for t_, c_ in zip((a.params, a.m, a.v), snap):
            t_.copy_(c_)
        a.refresh_shadows()
        ts.opt_step = step0
        ts.set_batch(latent.cuda(), noise.cuda(), t.cuda(), mask.cuda(), time_ids=tid.cuda() if xl else None, ids=[ids] * len(encs),
                     caption_token_lists=lists, ctx=ctx_c, pooled=pooled_c)
        ts.run(1e-3, lr_ti=0.0)
        torch.cuda.synchronize()
        dl, dp = abs(float(ts.loss) - l_ref) / abs(l_ref), float((a.params - p_ref).abs().max())
        assert ts._cond_cached and dl <= 1e-6 and dp <= 1e-7, f"cached conditioning: loss rel diff {dl}, max parameter diff {dp} (lr 1e-3)"


@pytest.mark.parametrize("version,B,rank,dora,w_tok", [("tiny15", 2, 16, False, 0.0), ("tinyxl", 2, 8, False, 0.0), ("tinyxl", 2, 16, True, 0.0), ("tiny15", 2, 24, True, 0.0),
                                                       ("tinyxl", 2, 16, False, 2e-3),        # + tok_cond_reg_w: second pass through the adapters, accumulating dA / dB launch
                                                       ("tinyxl", 2, 16, True, 2e-3)])        # ... through weight-decomposed adapters: the magnitude gradients accumulate too (round 6)
def test_text_encoder_lora_gpu_matches_oracle(version, B, rank, dora, w_tok):
    """a21: LoRA on q/k/v/out_proj of the text encoders (trainer/optimizer.py:157-202) on the HIP path - fused into the
    stacked q|k|v GEMM (N-grouped forward, K-grouped dX) and the out_proj GEMM - against autograd through Hugging Face
    CLIP with merged projections; then graph replays with all three optimizers live."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from torch.func import functional_call
    from oracle import loss_ref as L
    from oracle import unet_ref as U
    import sd_lora_trainer_amd.clip as clip_mod
    import sd_lora_trainer_amd.step as step_mod
    import sd_lora_trainer_amd.unet as unet_mod
    from sd_lora_trainer_amd import topology
    cfg = U.CONFIGS[version]
    xl = cfg["addition"]
    w_ta, w_std = 2e-2, 0.01
    h = 32 if xl else 16
    sd = U.init_unet_state(cfg, seed=0)
    lora = U.init_lora(cfg, 4, seed=1, b_std=0.05)
    hf = ([_hf("quick_gelu", False, 64, 1, 11), _hf("gelu", True, 64, 1, 12, proj=cfg["proj_class_in"] - 6 * cfg["addition_time_embed_dim"])]
          if xl else [_hf("quick_gelu", False, 64, 2, 11)])
    for m in hf:
        for p in m.parameters():
            p.data = p.data.to(torch.bfloat16).float()
    g = torch.Generator().manual_seed(3)
    latent = torch.randn(B, 4, h, h, generator=g) * cfg["scaling_factor"]
    noise = torch.randn(B, 4, h, h, generator=g)
    mask = (torch.rand(B, 1, h, h, generator=g) * 0.95 + 0.05).repeat(1, 4, 1, 1).contiguous()
    t = torch.tensor([10, 900][:B])
    tid = torch.tensor([[1024., 1024, 0, 0, 8. * h, 8. * h]] * B) if xl else None
    lists, ids = _captions(B)

    rt = unet_mod.Runtime("cuda:0", B)
    unet = unet_mod.UNet(rt, topology.CONFIGS[version], sd, lora_rank=4)
    unet.arena.load(lora)
    te_arena = unet_mod.LoraArena(rt, rank, 1.0, problems=[], dora=dora)        # dora: use_dora on the text-encoder adapters (optimizer.py:157-165)
    sds = [{k: v.detach() for k, v in m.state_dict().items()} for m in hf]
    prefixes = ["text_encoder.", "text_encoder_2."]
    kw = [dict(heads=1, act="quick_gelu", mode="penultimate", with_projection=False), dict(heads=1, act="gelu", mode="penultimate", with_projection=True)] \
        if xl else [dict(heads=2, act="quick_gelu", mode="last", with_projection=False)]
    encs = [clip_mod.ClipTextEncoder(rt, f"te{i + 1}", sds[i], n_train=NTOK, arena=te_arena, lora_prefix=prefixes[i], **k) for i, k in enumerate(kw)]
    te_arena.finalize()
    gl = torch.Generator().manual_seed(21)
    bf = lambda x: x.to(torch.bfloat16).float()  # noqa: E731  (the compute copies are bf16)
    te_lora = {e["name"]: (bf(torch.randn(rank, e["K"], generator=gl) / rank), bf(torch.randn(e["N"], rank, generator=gl) * 0.05)) for e in te_arena.entries}
    if dora:
        for i_, pre in enumerate(prefixes[:len(hf)]):
            for name, (A, Bm, *_) in list(te_lora.items()):
                if name.startswith(pre):
                    w = sds[i_][name[len(pre):] + ".weight"]
                    te_lora[name] = (A, Bm, (w + te_arena.scale * Bm @ A).norm(dim=1) * (1.0 + 0.05 * torch.randn(w.shape[0], generator=gl)))
    te_arena.load(te_lora)
    text = step_mod.TextStack(rt, encs, pool_mode="first_eos", eos_token_id=EOS, arena=te_arena)
    from tests.test_ti_step_cpu import BOS
    tok = list(TRAIN_IDS)
    caps = [[5, 6, 7] + tok, tok, [5, 6, 7] + tok + [8, 9] + tok, tok + [10] + tok]
    reg_ids = torch.full((4, 77), EOS, dtype=torch.int64)
    for r_, c_ in enumerate(caps):
        reg_ids[r_, 0] = BOS
        reg_ids[r_, 1:1 + len(c_)] = torch.tensor(c_)
    ts = step_mod.TrainStep(rt, unet, latent_hw=(h, h), snr_gamma=5.0, l1_penalty=0.0, weight_decay=0.0, text=text, n_tokens=NTOK,
                            token_attention_loss_w=w_ta, ti_std_loss_w=w_std, tok_cond_reg_w=w_tok,
                            reg_caption_ids=[reg_ids.cuda()] * len(encs) if w_tok else None)
    ts.set_batch(latent.cuda(), noise.cuda(), t.cuda(), mask.cuda(), time_ids=tid.cuda() if xl else None, ids=[ids] * len(encs),
                 caption_token_lists=lists)
    ts.forward_backward()
    torch.cuda.synchronize()

    te_params, names, outs, routs = [], [], [], []
    for i, m in enumerate(hf):
        over = {}
        for name, (A, Bm, *mag) in te_lora.items():
            if not name.startswith(prefixes[i]):
                continue
            A, Bm = A.clone().requires_grad_(True), Bm.clone().requires_grad_(True)
            te_params += [A, Bm]
            names.append(name)
            key = name[len(prefixes[i]):] + ".weight"
            merged = sds[i][key] + te_arena.scale * Bm @ A
            if dora:
                mg = mag[0].clone().requires_grad_(True)
                te_params.append(mg)
                merged = (mg / merged.norm(dim=1).detach())[:, None] * merged
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
    if w_tok:
        rctx = torch.cat([routs[0].hidden_states[-2], routs[1].hidden_states[-2]], dim=-1) if xl else routs[0].last_hidden_state
        tokreg, tok_norm = L.prompt_norm_loss(rctx, 34.5 if xl else 27.8)
        total = total + w_tok * tokreg
        assert abs(float(ts.tok_reg_norm) - float(tok_norm)) <= 1e-2 * float(tok_norm)
    grads = torch.autograd.grad(total, te_params + embs)
    assert abs(float(ts.loss) - float(img_loss)) <= 2e-2 * float(img_loss)
    got = te_arena.export("grads")
    got_flat = torch.cat([x.reshape(-1) for n in names for x in got[n]])
    ref_flat = torch.cat([x.reshape(-1) for x in grads[:len(te_params)]])
    cos, rel = _cos_rel(got_flat, ref_flat)
    assert cos >= 0.985 and rel <= 0.12, f"text-encoder LoRA grads cos {cos} rel {rel}"
    for got_r, ref in zip(ts.ti.grad_rows, [ge[-NTOK:] for ge in grads[len(te_params):]]):
        cos, rel = _cos_rel(got_r, ref)
        assert cos >= 0.985 and rel <= 0.12, f"TI row grads cos {cos} rel {rel}"
    ts.capture(warmup=1)
    te0 = te_arena.params.clone()
    losses = []
    for i in range(6):
        ts.run(1e-3, lr_ti=1e-3, lr_te=1e-3)
        losses.append(ts.total_loss())
    assert all(torch.isfinite(torch.tensor(losses))) and losses[-1] < losses[0], losses
    assert not torch.equal(te0, te_arena.params)
    ts.run(1e-3, lr_ti=0.0, lr_te=1e-3)          # no frozen fast path while the text encoders are LoRA-trained
    torch.cuda.synchronize()
    assert math.isfinite(ts.total_loss())


def test_token_warmup_gpu_matches_oracle():
    """a20 on the HIP path (bf16 text encoders) against the Hugging Face + autograd + torch.optim.AdamW loop."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from oracle import loss_ref as L
    from oracle import unet_ref as U
    import sd_lora_trainer_amd.clip as clip_mod
    import sd_lora_trainer_amd.step as step_mod
    import sd_lora_trainer_amd.unet as unet_mod
    from sd_lora_trainer_amd import topology
    from tests.test_ti_step_cpu import BOS
    version, B, h, steps, lr = "tinyxl", 2, 32, 4, 2e-3
    cfg = U.CONFIGS[version]
    hf = [_hf("quick_gelu", False, 64, 1, 11), _hf("gelu", True, 64, 1, 12, proj=cfg["proj_class_in"] - 6 * cfg["addition_time_embed_dim"])]
    for m in hf:
        for p in m.parameters():
            p.data = p.data.to(torch.bfloat16).float()
    prompt = torch.full((77,), EOS, dtype=torch.int64)
    prompt[:5] = torch.tensor([BOS] + TRAIN_IDS + [EOS])
    target = torch.full((77,), EOS, dtype=torch.int64)
    target[:6] = torch.tensor([BOS, 5, 17, 33, 41, EOS])
    rt = unet_mod.Runtime("cuda:0", B)
    unet = unet_mod.UNet(rt, topology.CONFIGS[version], U.init_unet_state(cfg, seed=0), lora_rank=4)
    sds = [{k: v.detach().clone() for k, v in m.state_dict().items()} for m in hf]
    encs = [clip_mod.ClipTextEncoder(rt, "te1", sds[0], heads=1, act="quick_gelu", mode="penultimate", with_projection=False, n_train=NTOK),
            clip_mod.ClipTextEncoder(rt, "te2", sds[1], heads=1, act="gelu", mode="penultimate", with_projection=True, n_train=NTOK)]
    text = step_mod.TextStack(rt, encs, pool_mode="first_eos", eos_token_id=EOS)
    ts = step_mod.TrainStep(rt, unet, latent_hw=(h, h), text=text, n_tokens=NTOK)
    rows0 = [r.clone() for r in ts.ti.rows]
    got = ts.token_warmup([prompt] * 2, [target] * 2, steps, lr)

    embs = [m.get_input_embeddings().weight for m in hf]
    stats = [L.DistributionStats(e.detach()[:-NTOK].clone()) for e in embs]
    opt = torch.optim.AdamW(embs, lr=lr, weight_decay=0.0)

    def encode(ids):
        outs = [m(input_ids=ids.view(1, 77), output_hidden_states=True) for m in hf]
        return torch.cat([outs[0].hidden_states[-2], outs[1].hidden_states[-2]], dim=-1), outs[1].text_embeds
    with torch.no_grad():
        tgt, tgt_pooled = encode(target)
    ref = []
    for _ in range(steps):
        pe, pooled = encode(prompt)
        loss = 0.2 * L.target_prompt_loss(pe, tgt, pooled, tgt_pooled) + 0.5 * torch.stack([st.std_loss(e[-NTOK:]) for st, e in zip(stats, embs)]).mean()
        opt.zero_grad()
        loss.backward()
        for e in embs:
            e.grad.data[:-NTOK] *= 0.0
        opt.step()
        ref.append(float(loss))
    assert all(abs(a - b) <= 2e-2 * abs(b) for a, b in zip(got, ref)), (got, ref)
    assert got[-1] < got[0]
    for r, r0, e in zip(ts.ti.rows, rows0, embs):
        cos, rel = _cos_rel(r - r0, e.detach()[-NTOK:] - r0.cpu())
        assert cos >= 0.9, f"row update direction cos {cos}"        # Adam's first steps are ~lr*sign(g): sign flips of tiny g


@pytest.mark.parametrize("version", ["tiny15", "tinyxl"])
def test_optional_regularisers_gpu_match_oracle(version):
    """cond_reg_w, tok_cond_reg_w (second, 4-caption pass of the text encoders) and tok_cov_reg_w on the HIP path (bf16 activations)
    against the fp32 autograd oracle: loss values and the token-row gradients."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from sd_lora_trainer_amd import ops
    from tests.test_ti_step_cpu import run_optional_regularisers
    run_optional_regularisers(version, "cuda:0", ops, torch.bfloat16, rel_val=4e-3, rel_grad=4e-2)


@pytest.mark.parametrize("version,B,kinds,rank", [("tiny15", 2, ["tiny_l"], 4), ("tinyxl", 1, ["tiny_l", "tiny_g"], 16), ("tinyxl", 2, ["tiny_l", "tiny_g"], 24)])
def test_dora_step_and_trajectory_gpu(version, B, kinds, rank):
    """use_dora (peft weight-decomposed adapters, optimizer.py:86-95) on the HIP path: first step (prediction, A / B / magnitude
    gradients, token rows) and a 6-step AdamW trajectory under hipGraph replay against the fp32 oracle."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from oracle import unet_ref as U
    from tests.test_real_topology_gpu import TOL_BF16, _bf16_exact, run_step_and_trajectory
    sd = _bf16_exact(U.init_unet_state(U.CONFIGS[version], seed=0))
    run_step_and_trajectory(version, B, 32 if U.CONFIGS[version]["addition"] else 16, sd, kinds, device="cuda:0", tol=TOL_BF16, rank=rank, n_steps=6, dora=True)


def test_ti_step_gpu_with_fused_geglu_epilogues(monkeypatch):
    """The GEGLU epilogues of the feed-forward GEMMs (sdlt_gemm_params.epi_op 1 / 2) are off by default (slower than GEMM +
    element-wise kernel on the final kernels); SDLT_GEGLU_MIN_C opts in - the whole step with them on, against the same oracle."""
    monkeypatch.setenv("SDLT_GEGLU_MIN_C", "0")
    test_ti_step_gpu_matches_oracle("tinyxl", 2, False)
