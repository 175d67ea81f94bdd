"""Hot-loop timing of the text-encoder products: sdlt_strip_gemm vs the tiled sdlt_gemm_bf16 on the same shapes (hipGraph of 64 launches over
rotating weights, so every launch streams its weights from HBM like in the step).  python tools/strip_probe.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sd_lora_trainer_amd import ops as O

BF = torch.bfloat16
dev = "cuda"
SHAPES = [(3840, 1280, "ln"), (3840, 1280, "plain"), (1280, 1280, "res"), (5120, 1280, "ln_act"), (5120, 1280, "plain"), (1280, 5120, "res"), (5120, 1280, "dact"),
          (1280, 3840, "plain"), (2304, 768, "ln"), (2304, 768, "plain"), (768, 768, "res"), (3072, 768, "ln_act"), (768, 3072, "res")]
NROT = 16        # (per shape raised below so that the rotating weights exceed the 256 MB Infinity Cache, like the step's 5 GB of weights)


def bench(fn, n=64, reps=5):
    for i in range(4):
        fn(i)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(n):
            fn(i)
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    return best


for N, K, mode in SHAPES:
    M = 128
    x = torch.randn(M, K, device=dev).to(BF)
    NROT = max(16, (600 << 20) // (N * K * 2))
    ws = [(torch.randn(N, K, device=dev) * K ** -0.5).to(BF) for _ in range(NROT)]
    bias = torch.randn(N, device=dev)
    res = torch.randn(M, N, device=dev).to(BF)
    pre = torch.randn(M, N, device=dev).to(BF)
    c1, c2 = torch.randn(N, device=dev), torch.randn(N, device=dev)
    out, out2 = torch.zeros(M, N, device=dev, dtype=BF), torch.zeros(M, N, device=dev, dtype=BF)
    st = torch.zeros(M * 2, device=dev)
    xn = torch.zeros(M, K, device=dev, dtype=BF)
    gamma, beta = torch.ones(K, device=dev), torch.zeros(K, device=dev)
    kw = dict(B=1, T=77, Tp=128)

    def strip(i):
        w = ws[i % NROT]
        if mode == "ln":
            O.strip_gemm(x, w, out, ln=(c1, c2, 1e-5), stats=st, **kw)
        elif mode == "ln_act":
            O.strip_gemm(x, w, out, ln=(c1, c2, 1e-5), stats=st, act_out=("gelu", out2), **kw)
        elif mode == "res":
            O.strip_gemm(x, w, out, bias=bias, residual=res, **kw)
        elif mode == "dact":
            O.strip_gemm(x, w, out, dact_in=("gelu", pre), **kw)
        else:
            O.strip_gemm(x, w, out, **kw)

    def tiled(i):
        w = ws[i % NROT]
        if mode in ("ln", "ln_act"):
            O.layernorm_fwd(x, xn, st, gamma=gamma, beta=beta)
            O.gemm(xn, w, out, bias=bias, act_out=("gelu", out2) if mode == "ln_act" else None)
        elif mode == "res":
            O.gemm(x, w, out, bias=bias, residual=res)
        elif mode == "dact":
            O.gemm(x, w, out, dact_in=("gelu", pre))
        else:
            O.gemm(x, w, out)

    os.environ.pop("X", None)
    O.STRIP_SPLITK = 1
    t1 = bench(strip)
    O.STRIP_SPLITK = -1
    ts, tt = bench(strip), bench(tiled)
    mb = N * K * 2 / 1e6
    print(f"N{N:5d} K{K:5d} {mode:7s}  S=1 {t1:6.2f} us  S={O.strip_splitk(N, K, 1)} strip {ts:6.2f} us ({mb / ts * 1e3:6.0f} GB/s of weights)   tiled{'+LN' if mode.startswith('ln') else '   '} {tt:6.2f} us", flush=True)
