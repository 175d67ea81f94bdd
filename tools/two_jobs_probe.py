"""Two independent LoRA jobs in ONE process on one GPU: each with its own UNet / step / hipGraph, replayed on two streams.
Question: do two graph replays on two streams overlap (latency-bound kernels of one job filling the gaps of the other)?
python tools/two_jobs_probe.py"""
import sys, time
import torch
sys.path.insert(0, ".")
import bench                                    # noqa: E402
import sd_lora_trainer_amd.step as S            # noqa: E402
import sd_lora_trainer_amd.unet as M            # noqa: E402
from sd_lora_trainer_amd import topology        # noqa: E402

dev = torch.device("cuda:0")
cfg = topology.CONFIGS["sdxl"]
B, h = 1, 128


def make(seed, stream):
    with torch.cuda.stream(stream):
        rt = M.Runtime(dev, B)
        unet = M.UNet(rt, cfg, bench.make_state(cfg, dev, seed=seed), lora_rank=16)
        g = torch.Generator(device=dev).manual_seed(seed)
        for e in unet.arena.entries:
            e["A"].copy_(torch.randn(e["A"].shape, generator=g, device=dev) / 16)
        unet.arena.refresh_shadows()
        ts = S.TrainStep(rt, unet, latent_hw=(h, h))
        rn = lambda *s: torch.randn(*s, generator=g, device=dev)  # noqa: E731
        ts.set_batch(rn(B, 4, h, h) * 0.13, rn(B, 4, h, h), torch.randint(0, 1000, (B,), generator=g, device=dev), torch.ones(B, 4, h, h, device=dev),
                     rn(B, 77, 2048), rn(B, 1280), torch.tensor([[1024., 1024, 0, 0, 1024, 1024]], device=dev))
        ts.capture(warmup=2)
    stream.synchronize()
    return ts


s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
t1, t2 = make(1, s1), make(2, s2)


def timed(fn, n=20):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def one():
    with torch.cuda.stream(s1):
        t1.run(1e-4)


def both():
    with torch.cuda.stream(s1):
        t1.run(1e-4)
    with torch.cuda.stream(s2):
        t2.run(1e-4)


a = timed(one)
b = timed(both)
print(f"one job: {a:.2f} ms/step ({1e3 / a:.1f} img/s); two jobs on two streams: {b:.2f} ms per pair ({2e3 / b:.1f} img/s)")
