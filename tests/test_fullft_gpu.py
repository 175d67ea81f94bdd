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
