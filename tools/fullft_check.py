"""Full-size consistency check of the full fine-tune on the GPU: gradient arena from the deferred / batched weight-gradient plan
vs the layer-by-layer path, SDXL 512 px batch 4.  python tools/fullft_check.py"""
import sys
import torch
sys.path.insert(0, ".")
import bench                                    # noqa: E402  (make_state)
import sd_lora_trainer_amd.step as S            # noqa: E402
import sd_lora_trainer_amd.unet as M            # noqa: E402
from sd_lora_trainer_amd import fullft, topology   # noqa: E402

dev = torch.device("cuda:0")
cfg = topology.CONFIGS["sdxl"]
B, h = 4, 64
rt = M.Runtime(dev, B)
tr = fullft.WeightTrainer(rt)
unet = M.UNet(rt, cfg, bench.make_state(cfg, dev, seed=0), trainer=tr)
ts = S.TrainStep(rt, unet, latent_hw=(h, h))
g = torch.Generator(device=dev).manual_seed(1)
rn = lambda *s: torch.randn(*s, generator=g, device=dev)  # noqa: E731
ts.set_batch(rn(B, 4, h, h) * 0.13, rn(B, 4, h, h), torch.randint(0, 1000, (B,), generator=g, device=dev), torch.ones(B, 4, h, h, device=dev),
             rn(B, 77, 2048), rn(B, 1280), torch.tensor([[1024., 1024, 0, 0, 512, 512]] * B, device=dev))
ts.forward_backward()
torch.cuda.synchronize()
g_batched = tr.grads.clone()
tr.defer = False
tr.grads.zero_()
ts.forward_backward()
torch.cuda.synchronize()
a, b = g_batched, tr.grads                 # 2.57 G elements: reduce in chunks (BLAS dot is limited to 2^31 elements)
dot = na = nb = nd = 0.0
for i in range(0, a.numel(), 1 << 28):
    x, y = a[i:i + (1 << 28)].double(), b[i:i + (1 << 28)].double()
    dot += float((x * y).sum()); na += float((x * x).sum()); nb += float((y * y).sum()); nd += float(((x - y) ** 2).sum())
print("params", tr.n, "finite", bool(torch.isfinite(a).all()), "cos", dot / (na * nb) ** 0.5, "rel", (nd / nb) ** 0.5, "max|g|", float(b.abs().max()))
