"""Tile / split-K sweep for the 8x8-resolution convolutions of the SD1.5 step at batch 4 (M = 256 output pixels, K = 9 * Cin up to 23040):
auto vs every (tile, splitk) in a hipGraph hot loop with rotating weights (tools/gemm_probe.bench)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gemm_probe import bench
from sd_lora_trainer_amd import ops

shapes = [("conv 1280 8x8 b4", 256, 1280, 11520, False, ops.ConvGeom(4, 8, 8, 1280, 8, 8)), ("conv 1280 8x8 b4 lora", 256, 1280, 11520, True, ops.ConvGeom(4, 8, 8, 1280, 8, 8)),
          ("conv 2560->1280 8x8 b4", 256, 1280, 23040, False, ops.ConvGeom(4, 8, 8, 2560, 8, 8)), ("conv 1280 16x16 b4", 1024, 1280, 11520, False, ops.ConvGeom(4, 16, 16, 1280, 16, 16)),
          ("lin M256 1280x1280", 256, 1280, 1280, True, None), ("lin M256 ff1", 256, 10240, 1280, False, None), ("lin M256 ff2", 256, 1280, 5120, False, None)]
for (name, M, N, K, lora, conv) in shapes:
    res = []
    for tile in (1, 2, 3, 5, 8):
        for sk in (1, 2, 3, 4, 6, 8, 10, 13, 16):
            if sk > 1 and (K // 64) // sk < 3:
                continue
            try:
                us = bench(M, N, K, tile, sk, lora, conv)
            except Exception:
                continue
            res.append((us, tile, sk))
    res.sort()
    auto = bench(M, N, K, 0, 0, lora, conv)
    fl = 2.0 * M * N * K
    print(f"{name:24s} M{M} N{N} K{K} lora{int(lora)}: auto {auto:.1f}us ({fl / auto / 1e6:.0f} TF) | best " + ", ".join(f"t{t}/sk{q}:{u:.1f}" for u, t, q in res[:6]), flush=True)
