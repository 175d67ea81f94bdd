"""Prodigy (a16/a17 with `unet_optimizer_type` / `ti_optimizer` = "prodigy"): the oracle restatement of prodigyopt 1.0
(oracle/prodigy_ref.py, parity unpinned - the package is not available) checked through known answers and invariants, the
flat-arena formulation of the kernel (tests/emu_ops.prodigy_step mirrors csrc/optim.hip stage by stage) against it, and the
TrainStep / OptimizerCollection plumbing on the CPU emulation."""
import math

import pytest
import torch

from oracle import prodigy_ref as P
from oracle import unet_ref as U
from sd_lora_trainer_amd import step as step_mod
from sd_lora_trainer_amd import topology
from sd_lora_trainer_amd import unet as unet_mod
from sd_lora_trainer_amd.optimizer import OptimizerCollection, get_current_lr

from . import emu_ops

REF_KW = dict(lr=1.0, betas=(0.9, 0.99), decouple=True, use_bias_correction=True, safeguard_warmup=True)   # optimizer.py:24-34


def _grads(shapes, steps, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return [[torch.randn(s, generator=g) * scale for s in shapes] for _ in range(steps)]


def test_first_step_closed_form():
    """k = 0, d = d0: exp_avg = d0 (1-b1) g, exp_avg_sq = d0^2 (1-b2) g^2, numerator = 0 (p == p0) => d stays d0 and
    p moves by -dlr * (1-b1) g / (sqrt(1-b2) |g| + eps), dlr = d0 * sqrt(1-b2)/(1-b1)."""
    p = torch.tensor([0.5, -2.0, 3.0])
    g = torch.tensor([0.1, -0.4, 2.0])
    opt = P.Prodigy([p.clone()], weight_decay=0.0, **REF_KW)
    opt.step([g])
    d0, b1, b2 = 1e-6, 0.9, 0.99
    dlr = d0 * math.sqrt(1 - b2) / (1 - b1)
    want = p - dlr * (d0 * (1 - b1) * g) / (torch.sqrt(d0 * d0 * (1 - b2) * g * g) + d0 * 1e-8)
    torch.testing.assert_close(opt.params[0], want, rtol=1e-6, atol=0)
    grp = opt.param_groups[0]
    assert grp["d"] == d0 and grp["k"] == 1 and grp["d_numerator"] == 0.0
    assert P.effective_lr(grp) == pytest.approx(d0 * math.sqrt(1 - b2 ** 2) / (1 - b1 ** 2))


def test_growth_clamp_lr0_and_scale_invariance():
    shapes = [(7, 5), (11,)]
    gs = _grads(shapes, 12, seed=0)
    p_init = [torch.randn(s, generator=torch.Generator().manual_seed(9)) for s in shapes]
    run = {}
    for name, gscale, growth in (("a", 1.0, 1.05), ("b", 64.0, 1.05), ("free", 1.0, float("inf"))):
        opt = P.Prodigy([p.clone() for p in p_init], weight_decay=0.004, growth_rate=growth, **REF_KW)
        ds = []
        for noise in gs:      # gradient of gscale * (|p - 1|^2 / 2 + <noise, p> / 10): consistent direction, so d has to grow
            opt.step([gscale * ((p - 1.0) + 0.1 * nz) for p, nz in zip(opt.params, noise)])
            ds.append(opt.param_groups[0]["d"])
        run[name] = (opt, ds)
    ds = run["a"][1]
    # growth clamp (the first move away from d0 jumps straight to d_hat, by design), and d does adapt
    assert all(b <= a * 1.05 * (1 + 1e-12) for a, b in zip(ds, ds[1:]) if a != 1e-6) and ds[-1] > 1.5 * ds[0]
    assert run["free"][1][-1] > ds[-1]
    # the iterates do not depend on the scale of the loss (up to eps): d adapts inversely to the gradient scale
    for a, b in zip(run["a"][0].params, run["b"][0].params):
        torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-9)
    # lr == 0 (UNet frozen before freeze_unet_before_completion_f, main.py:290-291): nothing changes, k does not advance
    opt = run["a"][0]
    before = ([p.clone() for p in opt.params], dict(opt.param_groups[0]))
    opt.param_groups[0]["lr"] = 0.0
    opt.step(gs[0])
    assert all(torch.equal(a, b) for a, b in zip(opt.params, before[0]))
    assert {k: v for k, v in opt.param_groups[0].items() if k not in ("lr", "params")} == \
           {k: v for k, v in before[1].items() if k not in ("lr", "params")}


def test_rows_only_equals_masked_tables():
    """optimizer.py:107-155 steps the whole token-embedding tables, main.py:368-371 zeroes every gradient row but the
    last n_tokens: rows with zero gradients contribute nothing to numerator / denominator and never move."""
    V, D, n = 40, 16, 3
    g = torch.Generator().manual_seed(1)
    tables = [torch.randn(V, D, generator=g), torch.randn(V, 2 * D, generator=g)]
    full = P.Prodigy([t.clone() for t in tables], weight_decay=0.0, **REF_KW)
    rows = P.Prodigy([t[-n:].clone() for t in tables], weight_decay=0.0, **REF_KW)
    for _ in range(8):
        gr = [torch.randn(t.shape, generator=g) for t in tables]
        for x in gr:
            x[:-n] = 0
        full.step(gr)
        rows.step([x[-n:] for x in gr])
    for tf, tr, t0 in zip(full.params, rows.params, tables):
        torch.testing.assert_close(tf[-n:], tr, rtol=1e-6, atol=0)
        assert torch.equal(tf[:-n], t0[:-n])
    assert full.param_groups[0]["d"] == pytest.approx(rows.param_groups[0]["d"], rel=1e-9)


@pytest.mark.parametrize("wd,growth,l1", [(0.004, 1.05, 0.0), (0.0, float("inf"), 0.0), (0.004, 1.02, 0.03)])
def test_flat_arena_formulation_matches_oracle(wd, growth, l1):
    """The kernel's formulation: one flat arena, fp32 scalars on the device, L1 subgradient folded into the gradient."""
    shapes = [(6, 4), (4, 6), (10,)]
    gs = _grads(shapes, 15, seed=2, scale=0.3)
    p_init = [torch.randn(s, generator=torch.Generator().manual_seed(5)) for s in shapes]
    n = sum(p.numel() for p in p_init)
    opt = P.Prodigy([p.clone() for p in p_init], weight_decay=wd, growth_rate=growth, d_coef=2.0, **REF_KW)
    p = torch.cat([x.reshape(-1) for x in p_init]).clone()
    p0, m, v, s = p.clone(), torch.zeros(n), torch.zeros(n), torch.zeros(n)
    state = torch.zeros(16)
    state[:3] = 1e-6
    acc = torch.zeros(2, dtype=torch.float64)
    for i, gr in enumerate(gs):
        lr = 0.0 if i == 4 else 1.0
        opt.param_groups[0]["lr"] = lr
        opt.step([g + l1 / n * torch.sign(q) for g, q in zip(gr, opt.params)])
        hyper = torch.tensor([lr, 0.9, 0.99, math.sqrt(0.99), 1e-8, wd, 2.0, growth, l1 / n, 1.0, 1.0, 1.0, 1.0])
        emu_ops.prodigy_step(p, torch.cat([x.reshape(-1) for x in gr]), p0, m, v, s, hyper, state, acc)
        assert float(state[8]) == (0.0 if i == 4 else 1.0)
    torch.testing.assert_close(p, torch.cat([x.reshape(-1) for x in opt.params]), rtol=2e-5, atol=1e-9)
    assert float(state[0]) == pytest.approx(opt.param_groups[0]["d"], rel=1e-4)
    assert int(state[6]) == opt.param_groups[0]["k"] == 14


def test_trainstep_prodigy_plumbing():
    """TrainStep(optimizer='prodigy') on the CPU emulation: LoRA arena trajectory == oracle Prodigy on the same gradients;
    the OptimizerCollection handle exposes the group keys `get_current_lr` reads."""
    version, B, rank, h = "tiny15", 1, 4, 16
    cfg = U.CONFIGS[version]
    sd = U.init_unet_state(cfg, seed=0)
    lora = U.init_lora(cfg, rank, seed=1, b_std=0.05)
    g = torch.Generator().manual_seed(3)
    rt = unet_mod.Runtime("cpu", B, act_dtype=torch.float32, ops=emu_ops)
    unet = unet_mod.UNet(rt, topology.CONFIGS[version], sd, lora_rank=rank)
    unet.arena.load(lora)
    ts = step_mod.TrainStep(rt, unet, latent_hw=(h, h), l1_penalty=0.03, weight_decay=0.004, optimizer="prodigy",
                            prodigy_d_coef=1.0, prodigy_growth_rate=1.05)
    a = unet.arena
    opt = P.Prodigy([a.params.clone()], weight_decay=0.004, growth_rate=1.05, **REF_KW)
    for i in range(3):
        latent = torch.randn(B, 4, h, h, generator=g) * cfg["scaling_factor"]
        noise = torch.randn(B, 4, h, h, generator=g)
        ts.set_batch(latent, noise, torch.tensor([100 + 300 * i]), torch.ones(B, 4, h, h), torch.randn(B, 77, cfg["cross_dim"], generator=g))
        lr = 5e-5 * (i + 1)
        ts.run(lr)
        opt.param_groups[0]["lr"] = lr
        opt.step([a.grads + 0.03 / a.n * torch.sign(opt.params[0])])
        torch.testing.assert_close(a.params, opt.params[0], rtol=1e-5, atol=1e-9)

    class Cfg:
        ti_lr, ti_weight_decay, lora_weight_decay = 1e-3, 0.0, 0.004
    oc = OptimizerCollection(ts, Cfg)
    hnd = oc.optimizers["unet"]
    hnd.param_groups[0]["lr"] = lr
    assert get_current_lr(hnd) == pytest.approx(P.effective_lr(opt.param_groups[0]), rel=1e-4)
    assert hnd.param_groups[0]["k"] == 3 and hnd.param_groups[0]["use_bias_correction"]
    with pytest.raises(NotImplementedError):
        step_mod.TrainStep(rt, unet, latent_hw=(h, h), optimizer="sgd")
