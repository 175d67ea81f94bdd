"""Scratch: phase clock stamps of the wave-split-K GEMM (library built with -DSDLT_WSK_TRACE: tools/lab_build_obj.sh wsk.hip wsktrace:"-DSDLT_WSK_TRACE"): workgroup 0 / thread 0
stamps clock64() at kernel start (0), ring prefill issued (1), the first six K steps' data arrived (2..7), main loop done (8), after the barrier (9), partial tiles in LDS (12),
adapter T reduced (10), epilogue done (11).  Weights rotate through 24 distinct matrices (as in the step: never L2-resident); SDLT_WSK_PACK=0/1 picks the row-major / packed kernels."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sd_lora_trainer_amd import ops, _lib
BF = torch.bfloat16
lib = _lib.load()
lib.sdlt_wsk_trace_read.argtypes = [ctypes.c_void_p]
NROT = 24
for (M, N, K, lora, res, gk) in [(1024, 1280, 1280, True, False, 0), (1024, 1280, 1280, True, True, 0), (1024, 1280, 3840, True, True, 1280), (1024, 1280, 5120, False, True, 0), (1024, 1280, 10240, False, False, 0)]:
    X = torch.randn(M, K, device="cuda").to(BF)
    Ws = [(torch.randn(N, K, device="cuda") * 0.02).to(BF) for _ in range(NROT)]
    Y = torch.empty(M, N, device="cuda", dtype=BF); R = torch.randn(M, N, device="cuda").to(BF) if res else None
    G = K // gk if gk else 1
    kws = [{} for _ in range(NROT)]
    if lora:
        kws = [dict(lora=((torch.randn(16, K, device="cuda") / 16).to(BF), (torch.randn(N, 16 * G, device="cuda") * 0.01).to(BF), 1.0, None), lora_group_k=gk) for _ in range(NROT)]
    for W in Ws:
        ops.wsk_mark_frozen(W)
    acc = None
    DOUBLE = os.environ.get("WSK_TRACE_DOUBLE", "0") != "0"      # two launches back to back (different weights), the stamps are the second one's
    for rep in range(3 * NROT):
        if DOUBLE:
            ops.gemm(X, Ws[(rep + 7) % NROT], Y, residual=R, **kws[(rep + 7) % NROT])
        ops.gemm(X, Ws[rep % NROT], Y, residual=R, **kws[rep % NROT])
        if rep >= NROT:
            torch.cuda.synchronize()
            out = (ctypes.c_longlong * 16)()
            lib.sdlt_wsk_trace_read(out)
            t = list(out)
            rel = [x - t[0] for x in t[:16]]
            acc = rel if acc is None else [a + b for a, b in zip(acc, rel)]
    n = 2 * NROT
    a = [round(x / n) for x in acc]
    ns = min(6, K // 256)
    print(f"M{M} N{N} K{K} lora={lora} gk={gk} res={res} pack={os.environ.get('SDLT_WSK_PACK', '1')} double={int(DOUBLE)}: tile known {a[13]}, preloads issued {a[14]}, prefill issued {a[1]}, steps arrive {a[2:2 + ns]}, loop end {a[8]}, barrier {a[9]}, partials in LDS {a[12]}, T {a[10]}, end {a[11]}  (clock64 ticks, mean of {n} launches)", flush=True)
