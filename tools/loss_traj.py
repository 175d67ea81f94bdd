"""Loss trajectory of the bench workload on a fixed batch (debug aid): python tools/loss_traj.py [--jobs 2] [--no-graph]"""
import sys, math, argparse
import torch
sys.path.insert(0, ".")
import bench                                    # noqa: E402
import sd_lora_trainer_amd.step as S            # noqa: E402
import sd_lora_trainer_amd.unet as M            # noqa: E402
from sd_lora_trainer_amd import topology, ops   # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--jobs", type=int, default=1)
ap.add_argument("--no-graph", action="store_true")
ap.add_argument("--hint", action="store_true")
ap.add_argument("--side-stream", action="store_true")
ap.add_argument("--steps", type=int, default=200)
a = ap.parse_args()
dev = torch.device("cuda:0")
cfg = topology.CONFIGS["sdxl"]
B, h = 1, 128
if a.hint:
    ops.set_throughput_hint(True)


def make(seed):
    rt = M.Runtime(dev, B)
    unet = M.UNet(rt, cfg, bench.make_state(cfg, dev, seed=seed), lora_rank=16)
    g = torch.Generator(device=dev).manual_seed(100 + seed)
    for e in unet.arena.entries:
        e["A"].copy_(torch.randn(e["A"].shape, generator=g, device=dev) / 16)
    unet.arena.refresh_shadows()
    ts = S.TrainStep(rt, unet, latent_hw=(h, h))
    rn = lambda *s: torch.randn(*s, generator=g, device=dev)  # noqa: E731
    ts.set_batch(rn(B, 4, h, h) * 0.13, rn(B, 4, h, h), torch.randint(0, 1000, (B,), generator=g, device=dev), torch.ones(B, 4, h, h, device=dev),
                 rn(B, 77, 2048), rn(B, 1280), torch.tensor([[1024., 1024, 0, 0, 1024, 1024]], device=dev))
    if not a.no_graph:
        ts.capture(warmup=2)
    return ts


streams = [torch.cuda.Stream() for _ in range(a.jobs)] if (a.jobs > 1 or a.side_stream) else [torch.cuda.current_stream()]
jobs = []
for j, st in enumerate(streams):
    with torch.cuda.stream(st):
        jobs.append(make(j))
    st.synchronize()
out = []
for i in range(a.steps):
    for ts, st in zip(jobs, streams):
        with torch.cuda.stream(st):
            ts.run(bench.lr_at(i, a.steps))
    if i % 25 == 0 or i == a.steps - 1:
        torch.cuda.synchronize()
        out.append((i, [round(float(ts.loss), 4) for ts in jobs]))
print(out)
