"""Tile / split-K sweep for the CLIP-L text-encoder GEMMs at batch 4 / 8 (M = 512 / 1024 rows, width 768) - hipGraph hot loop, rotating weights."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gemm_probe import bench

shapes = [("clip-l fc2 b4", 512, 768, 3072), ("clip-l fc1 b4", 512, 3072, 768), ("clip-l o b4", 512, 768, 768), ("clip-l qkv b4", 512, 2304, 768), ("clip-l qkv dX b4", 512, 768, 2304),
          ("clip-l fc2 b8", 1024, 768, 3072), ("clip-l o b8", 1024, 768, 768), ("unet M256 K2560", 256, 1280, 2560), ("unet M256 K3840", 256, 1280, 3840)]
for (name, M, N, K) in shapes:
    res = []
    for tile in (1, 2, 3, 5):
        for sk in (1, 2, 3, 4, 5, 6, 8, 11):
            if sk > 1 and (K // 64) // sk < 3:
                continue
            try:
                us = bench(M, N, K, tile, sk, False, None)
            except Exception:
                continue
            res.append((us, tile, sk))
    res.sort()
    auto = bench(M, N, K, 0, 0, False, None)
    fl = 2.0 * M * N * K
    print(f"{name:18s} M{M} N{N} K{K}: auto {auto:.1f}us ({fl / auto / 1e6:.0f} TF) | best " + ", ".join(f"t{t}/sk{q}:{u:.1f}" for u, t, q in res[:6]), flush=True)
