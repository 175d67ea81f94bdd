"""Split-K / tile sweep for the M = 128 text-encoder GEMMs of the SDXL step (CLIP-L width 768, bigG width 1280) - hipGraph hot loop, rotating
weights (tools/gemm_probe.bench)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gemm_probe import bench

shapes = [("bigG fc2", 128, 1280, 5120), ("bigG fc1", 128, 5120, 1280), ("bigG qkv", 128, 3840, 1280), ("bigG o", 128, 1280, 1280), ("bigG qkv dX", 128, 1280, 3840),
          ("L fc2", 128, 768, 3072), ("L fc1", 128, 3072, 768), ("L qkv", 128, 2304, 768), ("L o", 128, 768, 768), ("L qkv dX", 128, 768, 2304)]
for (name, M, N, K) in shapes:
    res = []
    for tile in (2, 3):
        for sk in (1, 2, 3, 4, 5, 6, 8, 10, 12, 16):
            if sk > 1 and (K // 64) // sk < 2:
                continue
            try:
                us = bench(M, N, K, tile, sk, False, None)
            except Exception:
                continue
            res.append((us, tile, sk))
    res.sort()
    auto = bench(M, N, K, 0, 0, False, None)
    fl = 2.0 * M * N * K
    print(f"{name:12s} M{M} N{N} K{K}: auto {auto:.1f}us ({fl / auto / 1e6:.0f} TF) | best " + ", ".join(f"t{t}/sk{q}:{u:.1f}" for u, t, q in res[:7]), flush=True)
