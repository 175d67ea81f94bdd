"""What a cross-launch weight prefetch could buy the text encoders' strips: the same launches with the weights rotating through 600 MB (HBM-cold, like the step: 5 GB of
weights per step), 150 MB (resident in the 256 MB Infinity Cache but not in the 32 MB of L2s) and 16 MB (L2-resident).  The cold -> Infinity-Cache difference is the
upper bound of a prefetch of launch k + 1's weights during launch k.  python tools/strip_cache_probe.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sd_lora_trainer_amd import ops as O

BF, dev = torch.bfloat16, "cuda"


def bench(fn, n=64, reps=7):
    for i in range(4):
        fn(i)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(n):
            fn(i)
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best


for N, K, mode in [(3840, 1280, "ln"), (1280, 1280, "res"), (5120, 1280, "ln_act"), (1280, 5120, "res"), (2304, 768, "ln"), (768, 3072, "res")]:
    M = 128
    x = torch.randn(M, K, device=dev).to(BF)
    bias, res = torch.randn(N, device=dev), torch.randn(M, N, device=dev).to(BF)
    c1, c2 = torch.randn(N, device=dev), torch.randn(N, device=dev)
    out, out2, st = torch.zeros(M, N, device=dev, dtype=BF), torch.zeros(M, N, device=dev, dtype=BF), torch.zeros(M * 2, device=dev)
    line = f"N{N:5d} K{K:5d} {mode:7s}"
    for mb in (600, 150, 16):
        nrot = max(2, (mb << 20) // (N * K * 2))
        ws = [(torch.randn(N, K, device=dev) * K ** -0.5).to(BF) for _ in range(nrot)]

        def strip(i):
            w = ws[i % nrot]
            if mode == "ln":
                O.strip_gemm(x, w, out, ln=(c1, c2, 1e-5), stats=st, B=1, T=77, Tp=128)
            elif mode == "ln_act":
                O.strip_gemm(x, w, out, ln=(c1, c2, 1e-5), stats=st, act_out=("gelu", out2), B=1, T=77, Tp=128)
            else:
                O.strip_gemm(x, w, out, bias=bias, residual=res, B=1, T=77, Tp=128)
        line += f"   {mb:4d} MB: {bench(strip, n=max(64, 2 * nrot)):6.2f} us"
        del ws
    print(line)
