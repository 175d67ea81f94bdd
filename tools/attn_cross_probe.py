"""Cross-attention backward (single-pass kernel): time vs number of query splits."""
import sys, os, math, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sd_lora_trainer_amd import ops
BF = torch.bfloat16
def timeit(fn, reps=20):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (5 * reps)
for (name, B, H, Nq, Nk, Nkp, d) in [("cross N1024 H20", 1, 20, 1024, 77, 128, 64), ("cross N4096 H10", 1, 10, 4096, 77, 128, 64)]:
    C = H * d
    r = lambda n: torch.randn(B * n, C, device="cuda").to(BF)
    Q, K, V, dO = r(Nq), r(Nkp), r(Nkp), r(Nq)
    O = torch.zeros(B * Nq, C, dtype=BF, device="cuda"); L = torch.zeros(B * H * Nq, device="cuda"); D = torch.zeros_like(L)
    dQ, dK, dV = torch.zeros_like(Q), torch.zeros_like(K), torch.zeros_like(V)
    kw = dict(B=B, H=H, Nq=Nq, Nk=Nk, Nqp=Nq, Nkp=Nkp, d=d, scale=1 / math.sqrt(d))
    out = []
    for qs in (2, 4, 8, 16, 32, 64):
        if qs > Nq // 64: continue
        extra = dict(qsplit=qs, dK32=torch.empty(qs * B * Nkp, C, device="cuda"), dV32=torch.empty(qs * B * Nkp, C, device="cuda"))
        out.append(f"qs{qs}:{timeit(lambda: ops.attn_bwd(Q, K, V, None, None, O, L, dO, None, D, dQ, dK, dV, **kw, **extra)):.1f}us")
    print(name, " ".join(out))
