"""Correctness + hot-loop timing of the wave-split-K GEMM lab kernel against ops.gemm on the M = 1024 / 4096 transformer shapes."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sd_lora_trainer_amd import _lib, ops as O

BF = torch.bfloat16
lib = _lib.load()
fn = lib.sdlt_wsk_gemm_lab
fn.restype = C.c_int32
i32, i64, vp = C.c_int32, C.c_int64, C.c_void_p
fn.argtypes = [vp, i64, vp, i64, i32, i32, i32, vp, vp, i64, vp, i64, i32, vp]
NROT = 12


def wsk(x, w, y, bias, res, variant):
    rc = fn(x.data_ptr(), x.stride(0), w.data_ptr(), w.stride(0), x.shape[0], w.shape[0], w.shape[1], bias.data_ptr() if bias is not None else None,
            res.data_ptr() if res is not None else None, res.stride(0) if res is not None else 0, y.data_ptr(), y.stride(0), variant, torch.cuda.current_stream().cuda_stream)
    assert rc == 0, lib.sdlt_last_error()


def bench(f, n=48, reps=5):
    for i in range(3):
        f(i)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(n):
            f(i)
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best


for M, N, K in [(1024, 1280, 1280), (1024, 1280, 3840), (1024, 1280, 5120), (1024, 1280, 10240), (4096, 640, 2560), (4096, 640, 5120), (1024, 5120, 1280), (1024, 3840, 1280)]:
    x = torch.randn(M, K, device="cuda").to(BF)
    ws = [(torch.randn(N, K, device="cuda") * K ** -0.5).to(BF) for _ in range(NROT)]
    bias = torch.randn(N, device="cuda")
    res = torch.randn(M, N, device="cuda").to(BF)
    y, y2 = torch.zeros(M, N, device="cuda", dtype=BF), torch.zeros(M, N, device="cuda", dtype=BF)
    ref = (x.float() @ ws[0].float().t() + bias + res.float())
    line = f"M{M:5d} N{N:5d} K{K:6d}: "
    for v, name in ((0, "64x80"), (16, "64x80/2d"), (17, "64x64/2d")):
        tile_n = 80 if (v & 15) == 0 else 64
        if N % tile_n or (N // tile_n) % 8:
            continue
        wsk(x, ws[0], y, bias, res, v)
        err = float((y.float() - ref).abs().max() / ref.abs().max())
        t = bench(lambda i: wsk(x, ws[i % NROT], y, bias, res, v))
        line += f"wsk {name} {t:6.2f} us (err {err:.1e}, {2 * M * N * K / t * 1e-6:5.0f} TF/s)  "
    tt = bench(lambda i: O.gemm(x, ws[i % NROT], y2, bias=bias, residual=res))
    print(line + f"tiled {tt:6.2f} us ({2 * M * N * K / tt * 1e-6:5.0f} TF/s)", flush=True)
