"""Host-logic test (CPU, no GPU): the explicit forward/backward plan of sd-lora-trainer_amd/unet.py + step.py,
driven through the torch emulation of the C-ABI ops (tests/emu_ops.py) in fp32, must reproduce the oracle's
autograd (oracle/unet_ref.py + loss_ref.py) - prediction, loss, every LoRA gradient, the gradient w.r.t. the
text conditioning and one AdamW step."""
import pytest
import torch

from oracle import loss_ref as L
from oracle import unet_ref as U
from tests import emu_ops

import sd_lora_trainer_amd.step as step_mod
import sd_lora_trainer_amd.unet as unet_mod
from sd_lora_trainer_amd import topology


def test_topology_matches_oracle():
    for v in ("sd15", "sdxl", "tiny15", "tinyxl"):
        assert list(topology.param_shapes(topology.CONFIGS[v]).items()) == list(U.param_shapes(U.CONFIGS[v]).items())
        assert topology.lora_targets(topology.CONFIGS[v]) == U.lora_targets(U.CONFIGS[v])


def _oracle_step(cfg, sd, lora, rank, latent, noise, t, mask, ctx, add, gamma, l1w):
    params = []
    lora_g = {}
    for k, (A, B) in lora.items():
        A = A.clone().requires_grad_(True)
        B = B.clone().requires_grad_(True)
        lora_g[k] = (A, B)
        params += [A, B]
    ctx = ctx.clone().requires_grad_(True)
    acp = L.ddpm_alphas_cumprod()
    noisy = L.add_noise(acp, latent, noise, t)
    pred = U.unet_forward(cfg, sd, noisy, t, ctx, add, lora=lora_g, lora_scale=1.0)
    loss = L.diffusion_loss(pred, noise, noisy, mask, acp, t, snr_gamma=gamma)
    grads = torch.autograd.grad(loss, params + [ctx])
    return pred.detach(), loss.detach(), {k: (grads[2 * i], grads[2 * i + 1]) for i, k in enumerate(lora)}, grads[-1]


@pytest.mark.parametrize("fold", ["default", "fold-with-partials", "fold-k-walk-statistics", "fold-attention-only"])
@pytest.mark.parametrize("version,B,gamma", [("tiny15", 2, 5.0), ("tinyxl", 1, 0.0)])
def test_engine_matches_oracle_autograd(version, B, gamma, fold, monkeypatch):
    # the LayerNorm fold (unet.Linear.fold_ln) is on for 1280-wide blocks by default - never on the toy topologies; the other cases
    # switch it on at every width so that the host plumbing (folded operands, adapter constants, producer row partials / K-walk statistics /
    # the per-pass fallback of norm3 to the LayerNorm launch, the normalised rows written by the LayerNorm backward) is checked against autograd too
    if fold != "default":
        monkeypatch.setattr(unet_mod, "LN_FOLD_WIDTH", 32)
        monkeypatch.setattr(unet_mod, "PARTS", fold == "fold-with-partials")
        monkeypatch.setattr(unet_mod, "LN_FOLD", 3 if fold == "fold-attention-only" else 7)
    torch.manual_seed(0)
    cfg = U.CONFIGS[version]
    rank, h = 4, 16
    sd = U.init_unet_state(cfg, seed=0)
    lora = U.init_lora(cfg, rank, seed=1, b_std=0.05)
    g = torch.Generator().manual_seed(3)
    latent = torch.randn(B, 4, h, h, generator=g) * cfg["scaling_factor"]
    noise = torch.randn(B, 4, h, h, generator=g)
    mask = (torch.rand(B, 1, h, h, generator=g) * 0.95 + 0.05).repeat(1, 4, 1, 1)
    t = torch.tensor([10, 900][:B])
    ctx = torch.randn(B, 77, cfg["cross_dim"], generator=g)
    add = pooled = tid = None
    if cfg["addition"]:
        pooled = torch.randn(B, cfg["proj_class_in"] - 6 * cfg["addition_time_embed_dim"], generator=g)
        tid = torch.tensor([[1024., 1024, 0, 0, 128, 128]] * B)
        add = {"text_embeds": pooled, "time_ids": tid}
    pred_o, loss_o, grads_o, gctx_o = _oracle_step(cfg, sd, lora, rank, latent, noise, t, mask, ctx, add, gamma, 0.0)

    rt = unet_mod.Runtime("cpu", B, act_dtype=torch.float32, ops=emu_ops)
    rt.keep_daam_maps = True
    unet = unet_mod.UNet(rt, topology.CONFIGS[version], sd, lora_rank=rank)
    unet.arena.load(lora)
    n_folded = len(unet.arena.ln_items)
    assert (n_folded == 0) == (fold == "default"), n_folded      # (four adapters per block: q, k, v of attn1 and attn2.to_q)
    ts = step_mod.TrainStep(rt, unet, latent_hw=(h, h), snr_gamma=gamma, l1_penalty=0.0, weight_decay=0.0)
    ts.set_batch(latent, noise, t, mask, ctx, pooled, tid)
    pred = ts.forward_backward()
    pred = pred.reshape(B, h, h, 4).permute(0, 3, 1, 2)
    torch.testing.assert_close(pred, pred_o, rtol=1e-3, atol=1e-4)
    torch.testing.assert_close(ts.loss[0], loss_o, rtol=1e-4, atol=1e-6)
    got = unet.arena.export("grads")
    assert set(got) == set(grads_o)
    for k in grads_o:
        for a, b, nm in zip(got[k], grads_o[k], "AB"):
            scale = max(float(b.abs().max()), 1e-8)
            assert float((a - b).abs().max()) <= 2e-3 * scale + 1e-7, (k, nm, float((a - b).abs().max()), scale)
    gctx = ts.dctx.view(B, unet_mod.CTX_PAD, -1)
    torch.testing.assert_close(gctx[:, :77], gctx_o, rtol=2e-3, atol=1e-6 + 2e-3 * float(gctx_o.abs().max()))
    assert float(gctx[:, 77:].abs().max()) == 0.0
    # hooked DAAM score maps, reference hook order
    _, daam_o = U.unet_forward(cfg, sd, L.add_noise(L.ddpm_alphas_cumprod(), latent, noise, t), t, ctx, add,
                               lora={k: v for k, v in lora.items()}, return_daam=True)
    assert [n for n, _ in rt.daam] == [n for n, _ in daam_o]
    for (_, s), (_, so) in zip(rt.daam, daam_o):
        torch.testing.assert_close(s[:, :, :77], so, rtol=1e-3, atol=1e-3)
    # per-resolution sums of the hooked maps (what the token-attention loss consumes)
    for N, (ssum, nl, _) in rt.daam_sums.items():
        ref = sum(so for _, so in daam_o if so.shape[1] == N)
        assert nl == sum(1 for _, so in daam_o if so.shape[1] == N)
        torch.testing.assert_close(ssum.view(B, N, -1)[:, :, :77], ref, rtol=1e-3, atol=1e-3)

    # one AdamW step with L1 (main.py:353-356) against the oracle restatement
    ts.l1_penalty, ts.wd = 0.03, 0.004
    p0 = unet.arena.params.clone()
    g0 = unet.arena.grads.clone()
    ts.set_hyper(1e-3)
    ts.optimizer_step()
    n = p0.numel()
    gref = g0 + 0.03 * torch.sign(p0) / n
    pref, m, v = p0.clone(), torch.zeros_like(p0), torch.zeros_like(p0)
    L.adamw_step(pref, gref, m, v, 1, 1e-3, weight_decay=0.004)
    torch.testing.assert_close(unet.arena.params, pref, rtol=1e-5, atol=1e-7)


def test_gradient_accumulation():
    """main.py:362-366: loss / k per micro-step, optimizers step on every k-th micro-step (or on the last batch of an epoch).
    Accumulated gradient == mean of the micro-batch gradients; parameters only move at the boundary."""
    cfg, B, rank, h = U.CONFIGS["tiny15"], 1, 4, 16
    sd = U.init_unet_state(cfg, seed=0)
    lora = U.init_lora(cfg, rank, seed=1, b_std=0.05)
    g = torch.Generator().manual_seed(4)
    batches = [(torch.randn(B, 4, h, h, generator=g) * cfg["scaling_factor"], torch.randn(B, 4, h, h, generator=g), torch.tensor([100 + 400 * i]),
                torch.ones(B, 4, h, h), torch.randn(B, 77, cfg["cross_dim"], generator=g)) for i in range(3)]

    def make(ga):
        rt = unet_mod.Runtime("cpu", B, act_dtype=torch.float32, ops=emu_ops)
        unet = unet_mod.UNet(rt, topology.CONFIGS["tiny15"], sd, lora_rank=rank)
        unet.arena.load(lora)
        return unet, step_mod.TrainStep(rt, unet, latent_hw=(h, h), l1_penalty=0.0, weight_decay=0.0, grad_accum=ga)
    u1, t1 = make(1)
    gs = []
    for b in batches[:2]:
        t1.set_batch(*b)
        t1.forward_backward()
        gs.append(u1.arena.grads.clone())
    u2, t2 = make(2)
    p0 = u2.arena.params.clone()
    t2.set_batch(*batches[0])
    t2.run(1e-3)
    assert torch.equal(u2.arena.params, p0) and t2.opt_step == 0          # micro-step: no optimizer step
    t2.set_batch(*batches[1])
    t2.run(1e-3)
    assert t2.opt_step == 1 and not torch.equal(u2.arena.params, p0)
    torch.testing.assert_close(u2.arena.grads, 0.5 * (gs[0] + gs[1]), rtol=1e-5, atol=1e-9)
    pref, m, v = p0.clone(), torch.zeros_like(p0), torch.zeros_like(p0)
    L.adamw_step(pref, 0.5 * (gs[0] + gs[1]), m, v, 1, 1e-3, weight_decay=0.0)
    torch.testing.assert_close(u2.arena.params, pref, rtol=1e-5, atol=1e-8)
    # last batch of an epoch forces the step even though the accumulation window is not full
    p1 = u2.arena.params.clone()
    t2.set_batch(*batches[2])
    t2.run(1e-3, last_batch=True)
    assert t2.opt_step == 2 and not torch.equal(u2.arena.params, p1) and t2._micro == 0
