"""Hot-loop timing of the wave-split-K GEMM (sdlt_wsk_gemm through ops.gemm) against the tiled kernel on the M = 1024 transformer shapes, plain and with a
rank-16 adapter; SDLT_WSK_STAGGER=0/1 A/B (odd waves refill after their MFMAs)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sd_lora_trainer_amd import ops as O

BF = torch.bfloat16
NROT = 12


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


for M, N, K, lora, gk in [(1024, 1280, 1280, False, 0), (1024, 1280, 1280, True, 0), (1024, 1280, 3840, True, 1280), (1024, 1280, 3840, False, 0), (1024, 1280, 5120, False, 0), (1024, 1280, 10240, False, 0)]:
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
    line = f"M{M} N{N} K{K:6d} {'lora' if lora else 'plain'}{' gK' if gk else ''}: "
    for st in ("0", "1"):
        os.environ["SDLT_WSK_STAGGER"] = st
        O.WSK, O.WSK_LORA = True, True
        if not O.wsk_shape(M, N, K, lora):
            line += f"wsk(stagger {st}) n/a  "
            continue
        t = bench(lambda i: O.gemm(x, ws[i % NROT], y, **kw))
        line += f"wsk(stagger {st}) {t:6.2f} us ({2 * M * N * K / t * 1e-6:5.0f} TF/s)  "
    O.WSK = False
    tt = bench(lambda i: O.gemm(x, ws[i % NROT], y, **kw))
    O.WSK = True
    print(line + f"tiled {tt:6.2f} us ({2 * M * N * K / tt * 1e-6:5.0f} TF/s)", flush=True)
