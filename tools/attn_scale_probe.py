"""Self-attention N=1024, d=64: time vs number of heads (is the 320-workgroup case latency- or throughput-bound?)."""
import sys, os, math, torch
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
for H in (5, 10, 20, 40, 80):
    B, Nq, d = 1, 1024, 64
    C = H * d
    r = lambda n: torch.randn(B * n, C, device="cuda").to(BF)
    Q, K, V, dO = r(Nq), r(Nq), r(Nq), r(Nq)
    O = torch.zeros(B * Nq, C, dtype=BF, device="cuda"); L = torch.zeros(B * H * Nq, device="cuda"); D = torch.zeros_like(L)
    dQ, dK, dV = torch.zeros_like(Q), torch.zeros_like(K), torch.zeros_like(V)
    kw = dict(B=B, H=H, Nq=Nq, Nk=Nq, Nqp=Nq, Nkp=Nq, d=d, scale=1 / math.sqrt(d))
    tf = timeit(lambda: ops.attn_fwd(Q, K, V, None, O, L, **kw))
    tb = timeit(lambda: ops.attn_bwd(Q, K, V, None, None, O, L, dO, None, D, dQ, dK, dV, **kw))
    print(f"H={H:3d} ({16 * H} WGs): fwd {tf:6.1f} us  bwd {tb:6.1f} us")
