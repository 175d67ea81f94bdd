"""Re-check of the tile / split-K choice for the M = 1024 (32x32 tokens, C = 1280) classes that dominate the SDXL step's GEMM time
(hipGraph hot loop with rotating weights, tools/gemm_probe.bench): auto vs every (tile, splitk, stages)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gemm_probe import bench

shapes = [("ff2 fwd", 1024, 1280, 5120, False), ("ff1 dX", 1024, 1280, 10240, False), ("attn proj lora", 1024, 1280, 1280, True), ("attn proj", 1024, 1280, 1280, False),
          ("ff2 dX / ff1 half", 1024, 5120, 1280, False), ("qkv-width K", 1024, 1280, 3840, False), ("C640 proj lora", 4096, 640, 640, True), ("C640 ff2", 4096, 640, 2560, False),
          ("C640 ff1 dX", 4096, 640, 5120, False)]
for (name, M, N, K, lora) in shapes:
    res = []
    for tile in (1, 2, 3, 5, 8):
        for sk in (1, 2, 3, 4, 5, 6, 8):
            if sk > 1 and (K // 64) // sk < 4:
                continue
            for st in (0, 2):
                try:
                    us = bench(M, N, K, tile, sk, lora, None, stages=st)
                except Exception:
                    continue
                res.append((us, tile, st, sk))
    res.sort()
    auto = bench(M, N, K, 0, 0, lora, None)
    fl = 2.0 * M * N * K
    print(f"{name:18s} M{M} N{N} K{K} lora{int(lora)}: auto {auto:.1f}us ({fl / auto / 1e6:.0f} TF) | best " + ", ".join(f"t{t}/st{k}/sk{q}:{u:.1f}" for u, t, k, q in res[:6]), flush=True)
