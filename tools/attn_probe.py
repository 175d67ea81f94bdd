"""Scratch: attention kernel times for the SDXL shapes (graph-captured, 10 launches per replay)."""
import sys, os, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sd_lora_trainer_amd import ops
BF = torch.bfloat16
def timeit(fn, reps=10):
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
for (name, B, H, Nq, Nk, Nkp, d) in [("self N1024 H20", 1, 20, 1024, 1024, 1024, 64), ("self N4096 H10", 1, 10, 4096, 4096, 4096, 64),
                                     ("cross N1024 H20", 1, 20, 1024, 77, 80, 64), ("cross N4096 H10", 1, 10, 4096, 77, 80, 64),
                                     ("self N256 H20 (sdxl 512)", 4, 20, 256, 256, 256, 64), ("self N1024 H10 (sdxl 512)", 4, 10, 1024, 1024, 1024, 64),
                                     ("sd15 self N4096 H8 d40", 4, 8, 4096, 4096, 4096, 40)]:
    C = H * d
    r = lambda n: torch.randn(B * n, C, device="cuda").to(BF)
    Q, K, V, dO = r(Nq), r(Nkp), r(Nkp), r(Nq)
    Kt, Vt, Qt, dOt = K.t().contiguous(), V.t().contiguous(), Q.t().contiguous(), dO.t().contiguous()
    O = torch.zeros(B * Nq, C, dtype=BF, device="cuda"); L = torch.zeros(B * H * Nq, device="cuda"); D = torch.zeros_like(L)
    dQ, dK, dV = torch.zeros_like(Q), torch.zeros_like(K), torch.zeros_like(V)
    kw = dict(B=B, H=H, Nq=Nq, Nk=Nk, Nqp=Nq, Nkp=Nkp, d=d, scale=1 / math.sqrt(d))
    tf = timeit(lambda: ops.attn_fwd(Q, K, V, Vt, O, L, **kw))
    extra = {}
    if Nk < 128:
        qs = max(1, min((Nq + 63) // 64, 320 // (2 * H * B)))
        extra = dict(qsplit=qs, dK32=torch.empty(qs * B * Nkp, C, device="cuda"), dV32=torch.empty(qs * B * Nkp, C, device="cuda"))
    tb = timeit(lambda: ops.attn_bwd(Q, K, V, Kt, Qt, O, L, dO, dOt, D, dQ, dK, dV, **kw, **extra))
    fl = 4.0 * B * H * Nq * Nk * d
    print(f"{name:26s} fwd {tf:7.1f} us ({fl / tf / 1e6:6.0f} TF)   bwd {tb:7.1f} us ({2.5 * fl / tb / 1e6:6.0f} TF algorithmic 2.5x)")
