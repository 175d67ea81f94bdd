"""AdamW8bit without a GPU: the oracle's code books (structure; the product's table builder gives the same fp32 values), the oracle against AdamW with fp32
moments, and the host logic of `TrainStep(optimizer="AdamW8bit")` on the CPU emulation of the kernels."""
import math

import torch

from oracle import adam8bit_ref as A8


def test_code_books():
    from sd_lora_trainer_amd import ops
    for signed in (True, False):
        q = A8.create_dynamic_map(signed)
        assert q.dtype == torch.float32 and q.shape == (256,) and torch.equal(q, ops.dynamic_code_book(signed))
        assert bool((q[1:] > q[:-1]).all()) and float(q[-1]) == 1.0
        if signed:
            assert float(q[127]) == 0.0 and torch.equal(q[:127], -q[128:255].flip(0))
            pos = q[128:255]
        else:
            assert float(q[0]) == 0.0
            pos = q[1:255]
        # decade i holds 2^i (signed) / 2^(i+1) (unsigned) equally spaced values inside 10^(i-6) [0.1, 1]
        k = 0
        for i in range(7):
            n = 2 ** i if signed else 2 ** (i + 1)
            dec = pos[k: k + n].double() / 10.0 ** (i - 6)
            assert float(dec.min()) > 0.1 and float(dec.max()) < 1.0
            want = 0.1 + (torch.arange(n, dtype=torch.float64) + 0.5) * 0.9 / n
            assert float((dec - want).abs().max()) < 1e-6
            k += n
        assert k == len(pos)
    x = torch.tensor([-2.0, -1e-9, 0.0, 1e-9, 5.5e-7, 0.5, 0.9965, 2.0])
    q = A8.create_dynamic_map(True)
    idx = A8.nearest_code(x, q)
    assert idx.tolist()[0] == 0 and idx.tolist()[2] == 127 and idx.tolist()[4] == 128 and idx.tolist()[-1] == 255
    assert bool(((q[idx] - x).abs() <= (q[(idx - 1).clamp(0)] - x).abs()).all()) and bool(((q[idx] - x).abs() <= (q[(idx + 1).clamp(max=255)] - x).abs()).all())


def test_oracle_adamw8_tracks_fp32_adamw():
    gen = torch.Generator().manual_seed(5)
    rows, cols, lr, b1, b2, eps, wd = 70, 130, 1e-3, 0.9, 0.999, 1e-8, 0.01
    p0 = torch.randn(rows, cols, generator=gen) * 0.05
    st, p8 = A8.Adam8State(rows, cols), p0.clone()
    p32, m32, v32 = p0.clone(), torch.zeros(rows, cols), torch.zeros(rows, cols)
    signal = torch.randn(rows, cols, generator=gen) * 1e-3
    for step in range(1, 31):
        g = signal + torch.randn(rows, cols, generator=gen) * 2e-3
        m_unq = b1 * A8.moments(st)[0] + (1 - b1) * g
        p8 = A8.adamw8_step(p8, g, st, lr=lr, beta1=b1, beta2=b2, eps=eps, weight_decay=wd, step=step)
        assert bool(((A8.moments(st)[0] < 0) == (m_unq < 0)).all())      # the sign rule: a quantised first moment never changes side
        m32 = b1 * m32 + (1 - b1) * g
        v32 = b2 * v32 + (1 - b2) * g * g
        p32 = p32 * (1 - lr * wd) - lr / (1 - b1 ** step) * m32 / (v32.sqrt() / math.sqrt(1 - b2 ** step) + eps)
        if step == 1:          # the first step uses the unquantised moments: identical to AdamW up to the order of decay and step
            torch.testing.assert_close(p8, p32, rtol=1e-5, atol=1e-7)
    assert float((p8 - p32).norm() / (p32 - p0).norm()) <= 0.05
    m, v = A8.moments(st)
    assert float((m - m32).norm() / m32.norm()) <= 0.05 and float((v - v32).norm() / v32.norm()) <= 0.05


def test_trainstep_adamw8bit_host_logic():
    """Full fine-tune with `unet_optimizer_type: AdamW8bit` on the CPU emulation: byte moments for the matrices, fp32 for the vector region, the first step equals
    AdamW's, later steps stay close to the fp32-moment run."""
    from oracle import unet_ref as U

    from . import emu_ops
    from sd_lora_trainer_amd import fullft, step as step_mod, topology, unet as unet_mod
    cfg, h, B = U.CONFIGS["tiny15"], 16, 2
    gen = torch.Generator().manual_seed(3)
    runs = {}
    for opt in ("adamw", "AdamW8bit"):
        sd = U.init_unet_state(cfg, seed=0)          # (a fresh state per run: on the CPU the engine's fp32 views may alias the tensors it was built from)
        rt = unet_mod.Runtime("cpu", B, act_dtype=torch.float32, ops=emu_ops)
        tr = fullft.WeightTrainer(rt)
        unet = unet_mod.UNet(rt, topology.CONFIGS["tiny15"], sd, trainer=tr)
        ts = step_mod.TrainStep(rt, unet, latent_hw=(h, h), snr_gamma=5.0, optimizer=opt)
        assert ts.adam8 == (opt == "AdamW8bit")
        gen.manual_seed(3)
        traj = []
        for it in range(3):
            latent, noise = torch.randn(B, 4, h, h, generator=gen) * cfg["scaling_factor"], torch.randn(B, 4, h, h, generator=gen)
            t = torch.randint(0, 1000, (B,), generator=gen)
            ctx = torch.randn(B, 77, cfg["cross_dim"], generator=gen)
            ts.set_batch(latent, noise, t, torch.ones(B, 4, h, h), ctx, None, None)
            ts.forward_backward()
            ts.set_hyper(1e-4)
            ts.optimizer_step()
            traj.append(tr.params.clone())
        runs[opt] = (tr, traj)
    tr8, t8 = runs["AdamW8bit"]
    tr32, t32 = runs["adamw"]
    assert tr8.m is None and tr8.q8[0].dtype == torch.uint8 and tr8.q8[0].numel() == 4096 * tr8._plan.n_blocks and tr8.m_vec.numel() == tr8.nv
    assert int(tr8.q8[0].max()) > 0 and float(tr8.q8[2].max()) > 0
    torch.testing.assert_close(t8[0], t32[0], rtol=1e-5, atol=1e-7)
    d32 = float((t32[2] - t32[0]).norm())
    assert float((t8[2] - t32[2]).norm()) <= 0.1 * d32, (float((t8[2] - t32[2]).norm()), d32)
    assert len(tr8.opt_state()) == 5
