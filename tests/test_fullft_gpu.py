"""Full-UNet fine-tune on the MI355X (bf16 HIP path): every parameter's gradient against the fp32 oracle's autograd, then
hipGraph replays of the whole step (forward, backward incl. all weight gradients, AdamW over the arena, refresh of the bf16
operands) train a fixed batch.  Tolerances: bf16 activations and bf16 GEMM panels ~40 layers deep - whole-arena cosine >=
0.99 / relative L2 <= 8e-2 as for the LoRA gradients (tests/test_step_gpu.py); per tensor cosine >= 0.97."""
import pytest
import torch

from tests.test_fullft_cpu import _inputs, oracle_grads

pytestmark = pytest.mark.gpu


def _cos_rel(a, b):
    a, b = a.reshape(-1).double().cpu(), b.reshape(-1).double().cpu()
    return float(a @ b / (a.norm() * b.norm() + 1e-30)), float((a - b).norm() / (b.norm() + 1e-30))


@pytest.mark.parametrize("version,B", [("tiny15", 2), ("tinyxl", 2)])
def test_fullft_gpu_matches_oracle(version, B):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from oracle import unet_ref as U
    from sd_lora_trainer_amd import fullft, topology
    import sd_lora_trainer_amd.step as S
    import sd_lora_trainer_amd.unet as M
    cfg, h = U.CONFIGS[version], 16
    sd = {k: v.to(torch.bfloat16).float() for k, v in U.init_unet_state(cfg, seed=0).items()}     # both sides see bf16-exact weights
    latent, noise, mask, t, ctx, pooled, tid, add = _inputs(cfg, B, h)
    pred_o, loss_o, grads_o = oracle_grads(cfg, sd, latent, noise, t, mask, ctx, add)

    rt = M.Runtime("cuda:0", B)
    tr = fullft.WeightTrainer(rt)
    unet = M.UNet(rt, topology.CONFIGS[version], sd, trainer=tr)
    ts = S.TrainStep(rt, unet, latent_hw=(h, h), snr_gamma=5.0)
    dv = lambda x: x.cuda() if x is not None else None  # noqa: E731
    ts.set_batch(dv(latent), dv(noise), dv(t), dv(mask), dv(ctx), dv(pooled), dv(tid))
    pred = ts.forward_backward().float().cpu().reshape(B, h, h, 4).permute(0, 3, 1, 2)
    torch.cuda.synchronize()
    assert float((pred - pred_o).abs().max()) <= 4e-2 * float(pred_o.abs().max())
    assert abs(float(ts.loss) - loss_o) <= 2e-2 * abs(loss_o)
    got = tr.export("grads")
    names = list(grads_o)
    cos, rel = _cos_rel(torch.cat([got[k].reshape(-1) for k in names]), torch.cat([grads_o[k].reshape(-1) for k in names]))
    assert cos >= 0.99 and rel <= 8e-2, f"all-parameter gradient: cos {cos} rel {rel}"
    worst = min((_cos_rel(got[k], grads_o[k])[0], k) for k in names if grads_o[k].numel() >= 64)
    assert worst[0] >= 0.97, worst

    ts.capture(warmup=1)
    p0 = tr.params.clone()
    losses = []
    for i in range(8):
        ts.run(2e-4)
        losses.append(float(ts.loss))
    assert all(x == x for x in losses) and losses[-1] < losses[0], losses
    assert not torch.equal(p0, tr.params)
    # the bf16 operands follow the master: q projection of the first attention, both orientations
    att = next(a for a in unet.cross_attns)
    w = tr.view(att.to_q.went)
    assert torch.equal(att.to_q.W, w.to(torch.bfloat16)) and torch.equal(att.to_q.Wt, w.t().to(torch.bfloat16))


@pytest.mark.parametrize("version", ["tinyxl"])
def test_fullft_adamw8bit_step_follows_fp32_moments(version):
    """`unet_optimizer_type: AdamW8bit` through the whole captured step (TrainStep(optimizer="AdamW8bit"): sdlt_adamw8_shadow_refresh inside the graph) against the same
    step with fp32 moments: the first step moves the masters identically (the update uses the unquantised moments; only the order of step and decay differs), after 12
    steps on a fixed batch the two runs are within 8 % of the displacement, both train, and the operands the GEMMs read are the bf16 rounding of the 8-bit run's masters."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from oracle import unet_ref as U
    from sd_lora_trainer_amd import fullft, topology
    import sd_lora_trainer_amd.step as S
    import sd_lora_trainer_amd.unet as M
    cfg, h, B = U.CONFIGS[version], 16, 2
    latent, noise, mask, t, ctx, pooled, tid, add = _inputs(cfg, B, h)
    dv = lambda x: x.cuda() if x is not None else None  # noqa: E731
    runs = {}
    for opt in ("adamw", "AdamW8bit"):
        sd = {k: v.to(torch.bfloat16).float() for k, v in U.init_unet_state(cfg, seed=0).items()}
        rt = M.Runtime("cuda:0", B)
        tr = fullft.WeightTrainer(rt)
        unet = M.UNet(rt, topology.CONFIGS[version], sd, trainer=tr)
        ts = S.TrainStep(rt, unet, latent_hw=(h, h), snr_gamma=5.0, optimizer=opt)
        assert ts.adam8 == (opt == "AdamW8bit")
        ts.set_batch(dv(latent), dv(noise), dv(t), dv(mask), dv(ctx), dv(pooled), dv(tid))
        ts.capture(warmup=1)
        p0 = tr.params.clone()
        traj, losses = [], []
        for i in range(12):
            ts.run(2e-4)
            losses.append(float(ts.loss))
            if i in (0, 11):
                traj.append(tr.params.clone())
        runs[opt] = (tr, unet, p0, traj, losses)
    tr8, unet8, p0, t8, l8 = runs["AdamW8bit"]
    _, _, q0, t32, l32 = runs["adamw"]
    assert torch.equal(p0, q0) and tr8.m is None and tr8.q8[0].dtype == torch.uint8 and int(tr8.q8[0].max()) > 0
    d1 = float((t32[0] - q0).norm())
    assert float((t8[0] - t32[0]).norm()) <= 2e-3 * d1, (float((t8[0] - t32[0]).norm()), d1)
    d12 = float((t32[1] - q0).norm())
    assert float((t8[1] - t32[1]).norm()) <= 0.08 * d12, (float((t8[1] - t32[1]).norm()), d12)
    assert l8[-1] < l8[0] and l32[-1] < l32[0] and abs(l8[-1] - l32[-1]) <= 0.05 * abs(l32[0]), (l8, l32)
    att = next(a for a in unet8.cross_attns)
    w = tr8.view(att.to_q.went)
    assert torch.equal(att.to_q.W, w.to(torch.bfloat16)) and torch.equal(att.to_q.Wt, w.t().to(torch.bfloat16))
