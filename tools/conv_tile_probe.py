"""Hot-loop timing of the 3 x 3 convolutions of the 64 x 64 and 128 x 128 levels on the tiled implicit-GEMM kernel: the heuristic's choice (128 x 160 / 256 x 160 tiles with
split-K through fp32 slabs where the tile count is short of the chip) against 128 x 80 tiles WITHOUT a K split (tile 9, round 6: 4096 x 640 -> exactly 256 workgroups, each walks
the whole K = 9 Cin - no slabs, no last-arriver seam).  Weights rotate through > 256 MB; graph-replayed; results compared."""
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


for B, H, W, Cin, Cout, flip, lora in [(1, 64, 64, 640, 640, 0, False), (1, 64, 64, 640, 640, 1, False), (1, 64, 64, 640, 640, 0, True), (1, 64, 64, 1280, 640, 0, False),
                                       (1, 64, 64, 1920, 640, 0, False), (1, 64, 64, 960, 640, 0, False), (1, 64, 64, 640, 1280, 1, False), (1, 64, 64, 320, 640, 0, False),
                                       (1, 128, 128, 320, 320, 0, False), (1, 128, 128, 640, 320, 0, False), (1, 128, 128, 960, 320, 0, False), (4, 32, 32, 640, 640, 0, False)]:
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
    t0 = bench(lambda i: O.gemm(x, ws[i % NROT], y, **kw), n=NROT)
    y0 = y.clone()
    line = f"B{B} {H}x{W} Cin{Cin:5d} Cout{Cout:5d} flip{flip} {'lora' if lora else '    '}: heuristic {t0:7.2f} us ({2 * M * Cout * K / t0 * 1e-6:4.0f} TF/s)"
    for tile, sk in ((9, 1), (9, 2), (8, 1), (7, 1)):
        try:
            t1 = bench(lambda i: O.gemm(x, ws[i % NROT], y, tile=tile, splitk=sk, **kw), n=NROT)
            err = float((y.float() - y0.float()).abs().max() / y0.float().abs().max())
            line += f" | tile {tile} x{sk} {t1:7.2f} us ({100 * (t1 / t0 - 1):+5.1f} %, diff {err:.0e})"
        except Exception as e:
            line += f" | tile {tile} x{sk}: {str(e)[:40]}"
    print(line, flush=True)
    del ws
