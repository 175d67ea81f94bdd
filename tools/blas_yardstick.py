"""Yardstick, NOT a product path: what the vendor library (torch.matmul -> hipBLASLt / rocBLAS) needs for the plain products of the step's dominant
GEMM signatures, with the weights rotating through HBM like in the step (every layer has its own), next to this repo's kernels on the same
operands.  python tools/blas_yardstick.py"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sd_lora_trainer_amd import ops as O

BF = torch.bfloat16
shapes = [(1024, 1280, 1280), (1024, 1280, 5120), (1024, 1280, 10240), (1024, 10240, 1280), (1024, 3840, 1280), (1024, 5120, 1280),
          (4096, 640, 640), (4096, 640, 2560), (4096, 5120, 640), (4096, 1920, 640), (128, 1280, 2048)]
dev = "cuda"
for (M, N, K) in shapes:
    nW = max(4, min(160, int(600e6 / (N * K * 2))))       # > 256 MB of weights in rotation: no Infinity-Cache hits
    Ws = [torch.randn(N, K, device=dev, dtype=BF) * K ** -0.5 for _ in range(nW)]
    x = torch.randn(M, K, device=dev, dtype=BF)
    out = torch.empty(M, N, device=dev, dtype=BF)
    res = {}
    for name, fn in (("blas", lambda w: torch.matmul(x, w.t(), out=out)), ("sdlt", lambda w: O.gemm(x, w, out))):
        for w in Ws[:4]:
            fn(w)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for w in Ws:
                fn(w)
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            g.replay()
        e1.record(); torch.cuda.synchronize()
        res[name] = e0.elapsed_time(e1) * 1e3 / (5 * nW)
    fl = 2.0 * M * N * K
    print(f"M{M} N{N} K{K}: vendor {res['blas']:6.1f} us ({fl / res['blas'] / 1e6:5.0f} TF/s)   this repo {res['sdlt']:6.1f} us ({fl / res['sdlt'] / 1e6:5.0f} TF/s)   x{res['blas'] / res['sdlt']:.2f}")
