"""Yardstick, NOT a product path: the library attention (torch scaled_dot_product_attention, bf16, flash backend of this ROCm build) on the UNet's
self-attention shapes, forward and forward + backward, next to this repo's kernels.  python tools/attn_yardstick.py"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sd_lora_trainer_amd import ops as O

BF = torch.bfloat16


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g):
            for _ in range(n):
                fn()
        rep = g.replay
    except Exception:
        torch.cuda.synchronize()
        rep = lambda: [fn() for _ in range(n)]
    rep(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        rep()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (3 * n)


for (B, H, N, d) in [(1, 20, 1024, 64), (1, 10, 4096, 64), (4, 8, 4096, 40), (1, 20, 1024, 64)]:
    C = H * d
    q, k, v = (torch.randn(B, H, N, d, device="cuda", dtype=BF, requires_grad=True) for _ in range(3))
    do = torch.randn(B, H, N, d, device="cuda", dtype=BF)
    f_lib = timeit(lambda: torch.nn.functional.scaled_dot_product_attention(q, k, v))
    def fb():
        o = torch.nn.functional.scaled_dot_product_attention(q, k, v)
        torch.autograd.grad(o, (q, k, v), do)
    fb_lib = timeit(fb)
    # this repo: token-major [B*N, H*d] operands
    Q, K, V, dO = (torch.randn(B * N, C, device="cuda", dtype=BF) for _ in range(4))
    Oo, L = torch.empty(B * N, C, device="cuda", dtype=BF), torch.empty(B * H * N, device="cuda")
    D = torch.empty(B * H * N, device="cuda")
    dQ, dK, dV = (torch.empty(B * N, C, device="cuda", dtype=BF) for _ in range(3))
    kw = dict(B=B, H=H, Nq=N, Nk=N, Nqp=N, Nkp=N, d=d, scale=d ** -0.5)
    f_us = timeit(lambda: O.attn_fwd(Q, K, V, None, Oo, L, **kw))
    O.attn_fwd(Q, K, V, None, Oo, L, **kw)
    b_us = timeit(lambda: O.attn_bwd(Q, K, V, None, None, Oo, L, dO, None, D, dQ, dK, dV, **kw))
    fl = 4.0 * B * H * N * N * d
    print(f"B{B} H{H} N{N} d{d}: library fwd {f_lib:6.1f} us, fwd+bwd {fb_lib:6.1f} us   this repo fwd {f_us:6.1f} us ({fl / f_us / 1e6:4.0f} TF/s), bwd {b_us:6.1f} us, fwd+bwd {f_us + b_us:6.1f} us   x{fb_lib / (f_us + b_us):.2f}")
