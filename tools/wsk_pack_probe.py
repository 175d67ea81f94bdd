"""Hot-loop timing of the wave-split-K GEMM with the frozen weight in fragment-major order (registers) against the row-major form (LDS ring) on the M = 1024 shapes of the
1280-wide transformer blocks.  One process per setting of SDLT_WSK_WP_R / SDLT_WSK_STAGGER / SDLT_KERNEL_LIB (read once per process); weights rotate through > 256 MB."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sd_lora_trainer_amd import ops as O

BF = torch.bfloat16
NROT = 24


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


tag = f"R={os.environ.get('SDLT_WSK_WP_R', '3')} stagger={os.environ.get('SDLT_WSK_STAGGER', '1')} lib={os.path.basename(os.environ.get('SDLT_KERNEL_LIB', 'in-tree'))}"
pack = os.environ.get("SDLT_WSK_PACK", "1") != "0"
for M, N, K, lora, gk in [(1024, 1280, 1280, True, 0), (1024, 1280, 3840, True, 1280), (1024, 1280, 5120, False, 0), (1024, 1280, 10240, False, 0), (1024, 1280, 2560, False, 0)]:
    x = torch.randn(M, K, device="cuda").to(BF)
    ws = [(torch.randn(N, K, device="cuda") * K ** -0.5).to(BF) for _ in range(NROT)]
    bias = torch.randn(N, device="cuda")
    res = torch.randn(M, N, device="cuda").to(BF)
    G = K // gk if gk else 1
    A, Bu = torch.randn(16, K, device="cuda").to(BF) / 16, (torch.randn(N, 16 * G, device="cuda") * 0.05).to(BF)
    T = torch.zeros(M, 16 * G, device="cuda", dtype=BF)
    y = torch.zeros(M, N, device="cuda", dtype=BF)
    kw = dict(bias=bias, residual=res)
    if lora:
        kw.update(lora=(A, Bu, 1.0, T), lora_group_k=gk)
    assert O.wsk_shape(M, N, K, lora)
    t0 = bench(lambda i: O.gemm(x, ws[i % NROT], y, **kw))
    line = f"{tag}  M{M} N{N} K{K:6d} {'lora' if lora else 'plain'}{' gK' if gk else '   '}: row-major {t0:6.2f} us ({2 * M * N * K / t0 * 1e-6:4.0f} TF/s)"
    if pack:
        y0 = y.clone()
        for w in ws:
            O.wsk_mark_frozen(w)
            O.gemm(x, w, y, **kw)          # (the packed copy is made on the first eager use)
        t1 = bench(lambda i: O.gemm(x, ws[i % NROT], y, **kw))
        O.gemm(x, ws[(48 - 1) % NROT], y0, **kw)
        line += f"   packed {t1:6.2f} us ({2 * M * N * K / t1 * 1e-6:4.0f} TF/s)  {100 * (t1 / t0 - 1):+5.1f} %  bits {'same' if torch.equal(y, y0) else 'DIFFER'}"
    print(line, flush=True)
    del ws
