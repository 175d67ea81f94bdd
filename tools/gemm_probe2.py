import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemm_probe import bench
for (name, M, N, K, lora) in [("attn proj C1280", 1024, 1280, 1280, True), ("ff2", 1024, 1280, 5120, False), ("qkv fused-size", 1024, 3840, 1280, False), ("dn fused-size", 1024, 1280, 3840, False),
                              ("attn proj C640", 4096, 640, 640, True), ("qkv C640", 4096, 1920, 640, False), ("clip", 128, 1280, 1280, False), ("clip qkv", 128, 3840, 1280, False)]:
    res = []
    for tile in (1, 2, 3):
        for sk in (1, 3):
            for st in (0, 2):
                if sk > (K // 64) // 2: continue
                try: us = bench(M, N, K, tile, sk, lora, None, stages=st)
                except Exception as e: continue
                res.append((us, tile, sk, st))
    res.sort()
    print(f"{name:18s} M{M} N{N} K{K}: " + ", ".join(f"t{t}/s{k}/st{s}:{u:.1f}" for u, t, k, s in res[:8]))
