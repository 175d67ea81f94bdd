"""AdamW8bit (bitsandbytes 0.43.1 blockwise 8-bit Adam, trainer/optimizer.py:19-21) fused into the operand-refresh tiles: sdlt_adamw8_shadow_refresh through the
C-ABI against oracle/adam8bit_ref.py on the same inputs.  Quantisation is a step function: a moment that lands within an ulp of a code boundary may take the
neighbouring code on one side and not on the other, so codes are compared as "equal except for a 1e-3 fraction, never more than one step apart"."""
import math

import pytest
import torch

from oracle import adam8bit_ref as A8

pytestmark = pytest.mark.gpu
F32, BF16 = torch.float32, torch.bfloat16


def _hyper(dev, lr, b1, b2, eps, wd, step, gs=1.0):
    h = torch.zeros(16, dtype=F32, device=dev)
    h[:9] = torch.tensor([lr, b1, b2, eps, wd, 1 - b1 ** step, 1 - b2 ** step, 0.0, gs])
    return h


def _grad(g, shape, scale, kind):
    x = torch.randn(shape, generator=g) * scale
    if kind == "outliers":                       # a few entries 1e4 x the rest: most of the block sits in the code book's small decades
        x[torch.rand(shape, generator=g) < 2e-3] *= 1e4
    if kind == "sparse":                         # exact zeros (rows that never see a gradient) and whole zero blocks
        x[torch.rand(shape, generator=g) < 0.5] = 0.0
        x[: min(32, shape[0]), : min(64, shape[1])] = 0.0
    return x


CASES = [(64, 64, "plain"), (100, 77, "plain"), (33, 4096, "outliers"), (320, 2880, "sparse"), (1280, 1280, "outliers"), (5, 9, "plain"), (1, 8192, "plain")]


@pytest.mark.parametrize("rows,cols,kind", CASES)
def test_adamw8_tiles_match_oracle(rows, cols, kind):
    from sd_lora_trainer_amd import ops
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(rows * 7 + cols)
    lr, b1, b2, eps, wd = 1e-3, 0.9, 0.999, 1e-8, 0.01
    off = 128                                                     # the tensor sits inside a larger arena
    n = off + rows * cols + 64
    p = torch.zeros(n, dtype=F32, device=dev)
    p0 = torch.randn(rows, cols, generator=gen) * 0.05
    p[off: off + rows * cols] = p0.reshape(-1).to(dev)
    g = torch.zeros_like(p)
    W, Wt = torch.zeros(rows, (cols + 7) // 8 * 8, dtype=BF16, device=dev), torch.zeros(cols, (rows + 7) // 8 * 8, dtype=BF16, device=dev)
    plan = ops.ShadowPlan([(off, rows, cols, cols, W, Wt)], dev)
    tr_, tc_ = (rows + 63) // 64, (cols + 63) // 64
    m8, v8 = torch.zeros(4096 * plan.n_blocks, dtype=torch.uint8, device=dev), torch.zeros(4096 * plan.n_blocks, dtype=torch.uint8, device=dev)      # tile-major codes

    def to_tiles(mat):          # [rows, cols] -> [tiles][64][64] (positions outside the tensor: 0)
        full = torch.zeros(tr_ * 64, tc_ * 64, dtype=torch.uint8)
        full[:rows, :cols] = mat
        return full.view(tr_, 64, tc_, 64).permute(0, 2, 1, 3).reshape(-1)

    def from_tiles(t):
        return t.cpu().view(tr_, tc_, 64, 64).permute(0, 2, 1, 3).reshape(tr_ * 64, tc_ * 64)
    absmax = torch.zeros(4 * plan.n_blocks, dtype=F32, device=dev)
    tables = ops.q8_tables(dev)
    st = A8.Adam8State(rows, cols)
    pref = p0.clone()
    worst_frac, steps = 0.0, 6
    for step in range(1, steps + 1):
        gi = _grad(gen, (rows, cols), 1e-3 * (1 + step % 3), kind)
        g[off: off + rows * cols] = gi.reshape(-1).to(dev)
        # both sides start every step from the ORACLE's state, so one flipped code does not grow into a different trajectory
        m8.copy_(to_tiles(st.m8).to(dev))
        v8.copy_(to_tiles(st.v8).to(dev))
        absmax.copy_(st.tile_absmax().reshape(-1).to(dev))
        p[off: off + rows * cols] = pref.reshape(-1).to(dev)
        plan.adamw8(p, g, m8, v8, absmax, tables, _hyper(dev, lr, b1, b2, eps, wd, step))
        pref = A8.adamw8_step(pref, gi, st, lr=lr, beta1=b1, beta2=b2, eps=eps, weight_decay=wd, step=step)
        torch.cuda.synchronize()
        got_p = p[off: off + rows * cols].cpu().view(rows, cols)
        torch.testing.assert_close(got_p, pref, rtol=2e-6, atol=1e-7)          # (atol = 1e-4 of the largest possible update, lr: hardware rcp / sqrt in the kernel, cancellation in b1 m + (1 - b1) g)
        torch.testing.assert_close(absmax.cpu().view(-1, 4), st.tile_absmax(), rtol=2e-6, atol=0)
        for got, ref, name in ((m8, st.m8, "m8"), (v8, st.v8, "v8")):
            full = from_tiles(got)
            d = (full[:rows, :cols].int() - ref.int()).abs()
            assert int(d.max()) <= 1, (name, step, int(d.max()))
            worst_frac = max(worst_frac, float((d > 0).float().mean()))
            full[:rows, :cols] = 0
            assert int(full.max()) == 0, name          # positions of ragged tiles outside the tensor stay untouched
        # the compute copies are the bf16 rounding of the new masters, both orientations; nothing outside the tensor moved
        assert torch.equal(W[:, :cols].cpu(), got_p.to(BF16)) and torch.equal(Wt[:, :rows].cpu(), got_p.t().to(BF16))
        assert float(p[:off].abs().max()) == 0.0 and float(p[off + rows * cols:].abs().max()) == 0.0
    assert worst_frac <= 1e-3, worst_frac


def test_adamw8_follows_fp32_adamw():
    """Free-running: 40 steps of the kernel on its own state against (a) the oracle on ITS own state, (b) AdamW with fp32 moments.  The quantisation noise of the
    moments stays a few percent of the displacement (what bitsandbytes promises of its 8-bit optimizers)."""
    from sd_lora_trainer_amd import ops
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(11)
    rows, cols, lr, b1, b2, eps, wd = 192, 320, 1e-3, 0.9, 0.999, 1e-8, 0.01
    p0 = torch.randn(rows, cols, generator=gen) * 0.05
    p = p0.reshape(-1).clone().to(dev)
    g = torch.zeros_like(p)
    plan = ops.ShadowPlan([(0, rows, cols, cols, None, None)], dev)
    m8, v8 = torch.zeros(4096 * plan.n_blocks, dtype=torch.uint8, device=dev), torch.zeros(4096 * plan.n_blocks, dtype=torch.uint8, device=dev)
    absmax, tables = torch.zeros(4 * plan.n_blocks, dtype=F32, device=dev), ops.q8_tables(dev)
    st, pref = A8.Adam8State(rows, cols), p0.clone()
    p32, m32, v32 = p0.clone(), torch.zeros(rows, cols), torch.zeros(rows, cols)
    signal = torch.randn(rows, cols, generator=gen) * 1e-3          # a persistent direction under per-step noise, like a real gradient
    for step in range(1, 41):
        gi = signal + torch.randn(rows, cols, generator=gen) * 2e-3
        g.copy_(gi.reshape(-1).to(dev))
        plan.adamw8(p, g, m8, v8, absmax, tables, _hyper(dev, lr, b1, b2, eps, wd, step))
        pref = A8.adamw8_step(pref, gi, st, lr=lr, beta1=b1, beta2=b2, eps=eps, weight_decay=wd, step=step)
        m32 = b1 * m32 + (1 - b1) * gi
        v32 = b2 * v32 + (1 - b2) * gi * gi
        p32 = p32 * (1 - lr * wd) - lr / (1 - b1 ** step) * m32 / (v32.sqrt() / math.sqrt(1 - b2 ** step) + eps)
    got = p.cpu().view(rows, cols)
    disp = float((p32 - p0).norm())
    rel = lambda a, b: float((a - b).norm()) / disp  # noqa: E731
    assert rel(got, pref) <= 5e-3, rel(got, pref)                   # kernel vs oracle, each on its own state
    assert rel(got, p32) <= 0.05 and rel(pref, p32) <= 0.05, (rel(got, p32), rel(pref, p32))
    mo, vo = A8.moments(st)
    assert float((mo - m32).norm() / m32.norm()) <= 0.05 and float((vo - v32).norm() / v32.norm()) <= 0.05


@pytest.mark.parametrize("n", [2048 * 3, 2048 * 5 + 4 * 100, 4 * 7, 2048 * 64 + 1024])
def test_adamw8_flat_matches_oracle(n):
    """sdlt_adamw8_flat (the sharded optimizer's slices): blocks of 2048 consecutive elements, the last one short - against the oracle on the range viewed as a
    [n / 64, 64] matrix (a 32 x 64 block of it IS 2048 consecutive elements), every step from the oracle's state."""
    from sd_lora_trainer_amd import ops
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(n)
    lr, b1, b2, eps, wd = 1e-3, 0.9, 0.999, 1e-8, 0.01
    rows = (n + 63) // 64
    pad = rows * 64 - n
    p0 = torch.randn(n, generator=gen) * 0.05
    p, g = torch.zeros(n + 64, device=dev), torch.zeros(n + 64, device=dev)          # (16 floats of guard behind the range)
    m8, v8 = torch.zeros(n + 64, dtype=torch.uint8, device=dev), torch.zeros(n + 64, dtype=torch.uint8, device=dev)
    nb = (n + 2047) // 2048
    absmax, tables = torch.zeros(2 * nb, device=dev), ops.q8_tables(dev)
    st = A8.Adam8State(rows, 64)
    st.m8[-1, 64 - pad:] = 127          # the padding of the last row decodes to 0
    pref = torch.cat([p0, torch.zeros(pad)]).view(rows, 64)
    worst = 0.0
    for step in range(1, 6):
        gi = torch.cat([_grad(gen, (n,), 1e-3 * (1 + step % 3), "outliers" if step % 2 else "sparse".replace("sparse", "plain")), torch.zeros(pad)]).view(rows, 64)
        g[:n] = gi.reshape(-1)[:n].to(dev)
        p[:n] = pref.reshape(-1)[:n].to(dev)
        m8[:n] = st.m8.reshape(-1)[:n].to(dev)
        v8[:n] = st.v8.reshape(-1)[:n].to(dev)
        absmax.copy_(torch.stack([st.am[:, 0], st.av[:, 0]], 1).reshape(-1).to(dev))
        ops.adamw8_flat(p[:n], g[:n], m8[:n], v8[:n], absmax, tables, _hyper(dev, lr, b1, b2, eps, wd, step))
        pref = A8.adamw8_step(pref, gi, st, lr=lr, beta1=b1, beta2=b2, eps=eps, weight_decay=wd, step=step)
        torch.cuda.synchronize()
        torch.testing.assert_close(p[:n].cpu(), pref.reshape(-1)[:n], rtol=2e-6, atol=1e-7)
        torch.testing.assert_close(absmax.cpu().view(nb, 2), torch.stack([st.am[:, 0], st.av[:, 0]], 1), rtol=2e-6, atol=0)
        for got, ref in ((m8, st.m8), (v8, st.v8)):
            d = (got[:n].cpu().int() - ref.reshape(-1)[:n].int()).abs()
            assert int(d.max()) <= 1
            worst = max(worst, float((d > 0).float().mean()))
        assert float(p[n:].abs().max()) == 0.0 and int(m8[n:].max()) == 0 and int(v8[n:].max()) == 0          # nothing behind the range moved
    assert worst <= 2e-3, worst
