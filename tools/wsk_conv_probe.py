"""Hot-loop timing of the 3 x 3 convolutions of the 32 x 32 level: wave-split-K kernel (sdlt_wsk_conv, frozen weights packed) against the tiled implicit-GEMM kernel (split-K through
fp32 slabs).  Weights rotate through > 256 MB; graph-replayed."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sd_lora_trainer_amd import ops as O

BF = torch.bfloat16


def bench(f, n=24, reps=5):
    for i in range(n):
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


for B, H, W, Cin, Cout, flip, lora in [(1, 32, 32, 1280, 1280, 0, False), (1, 32, 32, 1280, 1280, 1, False), (1, 32, 32, 1280, 1280, 0, True), (1, 32, 32, 2560, 1280, 0, False),
                                       (1, 32, 32, 1920, 1280, 0, False), (1, 32, 32, 1280, 2560, 1, False), (1, 32, 32, 640, 1280, 0, False), (1, 32, 32, 1280, 640, 1, False),
                                       (4, 16, 16, 1280, 1280, 0, False)]:
    M, K = B * H * W, 9 * Cin
    NROT = max(6, (300 << 20) // (Cout * K * 2))
    x = torch.randn(M, Cin, device="cuda").to(BF)
    ws = [(torch.randn(Cout, K, device="cuda") * K ** -0.5).to(BF) for _ in range(NROT)]
    bias = torch.randn(Cout, device="cuda")
    res = torch.randn(M, Cout, device="cuda").to(BF)
    y = torch.zeros(M, Cout, device="cuda", dtype=BF)
    geom = O.ConvGeom(B, H, W, Cin, H, W, flip=flip)
    kw = dict(conv=geom, bias=bias, residual=res)
    if lora:
        kw.update(lora=((torch.randn(16, K, device="cuda") / 16).to(BF), (torch.randn(Cout, 16, device="cuda") * 0.05).to(BF), 1.0, torch.zeros(M, 16, device="cuda", dtype=BF)))
    ok = O.wsk_conv_shape(geom, Cout, 16 if lora else 0)
    O.WSK_CONV = False
    t0 = bench(lambda i: O.gemm(x, ws[i % NROT], y, **kw), n=NROT)
    y0 = y.clone()
    O.WSK_CONV = True
    line = f"B{B} {H}x{W} Cin{Cin:5d} Cout{Cout:5d} flip{flip} {'lora' if lora else '    '}: tiled {t0:7.2f} us ({2 * M * Cout * K / t0 * 1e-6:4.0f} TF/s)"
    if ok:
        t1 = bench(lambda i: O.gemm(x, ws[i % NROT], y, **kw), n=NROT)          # row-major weights
        for w in ws:
            O.wsk_mark_frozen(w)
        t2 = bench(lambda i: O.gemm(x, ws[i % NROT], y, **kw), n=NROT)          # packed
        err = float((y.float() - y0.float()).abs().max() / y0.float().abs().max())
        line += f"   wave-split-K {t1:7.2f} us   packed {t2:7.2f} us ({2 * M * Cout * K / t2 * 1e-6:4.0f} TF/s)  {100 * (t2 / t0 - 1):+5.1f} %   max rel diff {err:.1e}"
    else:
        line += "   (not a wave-split-K shape)"
    print(line, flush=True)
    del ws
