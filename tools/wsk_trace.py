"""Scratch: phase clock stamps of the wave-split-K GEMM (library built with -DSDLT_WSK_TRACE): workgroup 0 / thread 0 stamps clock64()
at kernel start (0), ring prefill issued (1), the first six K steps' data arrived (2..7), main loop done (8), after the barrier (9),
adapter T reduced (10), epilogue done (11)."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sd_lora_trainer_amd import ops, _lib
BF = torch.bfloat16
lib = _lib.load()
for (M, N, K, lora, res) in [(1024, 1280, 1280, True, False), (1024, 1280, 1280, True, True), (1024, 1280, 5120, False, True), (1024, 1280, 1280, False, False)]:
    X = torch.randn(M, K, device="cuda").to(BF); W = (torch.randn(N, K, device="cuda") * 0.02).to(BF)
    Y = torch.empty(M, N, device="cuda", dtype=BF); R = torch.randn(M, N, device="cuda").to(BF) if res else None
    kw = {}
    if lora:
        kw = dict(lora=((torch.randn(16, K, device="cuda") / 16).to(BF), (torch.randn(N, 16, device="cuda") * 0.01).to(BF), 1.0, None))
    for rep in range(5):
        ops.gemm(X, W, Y, residual=R, **kw)
        torch.cuda.synchronize()
    out = (ctypes.c_longlong * 16)()
    lib.sdlt_wsk_trace_read(out)
    t = list(out)
    rel = [x - t[0] for x in t[:12]]
    print(f"M{M} N{N} K{K} lora={lora} res={res}: start->prefill {rel[1]}, steps arrive {rel[2:2 + min(6, K // 256)]}, loop end {rel[8]}, barrier {rel[9]}, T {rel[10]}, end {rel[11]}")
