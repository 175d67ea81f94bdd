"""Scratch micro-benchmark: sdlt_gemm_bf16 kernel time vs (tile, splitk) for the SDXL shapes, measured inside a
captured hipGraph (20 launches per replay) so host launch overhead does not pollute small kernels."""
import sys, os, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sd_lora_trainer_amd import ops
BF = torch.bfloat16

def bench(M, N, K, tile, splitk, lora=False, conv=None, reps=20, stages=0):
    pad = int(os.environ.get('LDPAD', '0'))    # extra elements per row: probes L2/HBM channel camping of power-of-two-ish strides
    Kx = K if conv is None else conv.Cin
    X = torch.randn(M if conv is None else conv.B * conv.Hin * conv.Win, Kx + pad, device="cuda").to(BF)[:, :Kx]
    Ws = [torch.randn(N, K + pad, device="cuda").to(BF)[:, :K] for _ in range(int(os.environ.get('NW', '4')))]   # rotate weights: frozen weights are never L2-hot in the real step
    out = torch.empty(M, N, device="cuda", dtype=BF)
    if os.environ.get('HOT'):   # every row aliases row 0: all operand traffic hits in TCP/L2 (probe: is the K loop memory- or issue-bound?)
        X = X[:1].expand(X.shape[0], X.shape[1])
        Ws = [w[:1].expand(w.shape[0], w.shape[1]) for w in Ws]
    lo = None
    if lora:
        lo = (torch.randn(16, K, device="cuda").to(BF), torch.randn(N, 16, device="cuda").to(BF), 1.0, torch.empty(M, 16, device="cuda", dtype=BF))
    def run():
        for i in range(reps): ops.gemm(X, Ws[i % len(Ws)], out, lora=lo, conv=conv, tile=tile, splitk=splitk, stages=stages)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        run()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        run()
    gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): gr.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (5 * reps)

if __name__ == "__main__":
    shapes = [("attn out C1280", 1024, 1280, 1280, True, None), ("qkv C1280", 1024, 3840, 1280, False, None), ("ff1 C1280", 1024, 10240, 1280, False, None), ("ff2 C1280", 1024, 1280, 5120, False, None),
              ("ff1 bwd C1280", 1024, 1280, 10240, False, None), ("ff2 bwd", 1024, 5120, 1280, False, None),
              ("attn out C640", 4096, 640, 640, True, None), ("qkv C640", 4096, 1920, 640, False, None), ("ff1 C640", 4096, 5120, 640, False, None), ("ff2 C640", 4096, 640, 2560, False, None),
              ("ff1 bwd C640", 4096, 640, 5120, False, None), ("ff2 bwd C640", 4096, 2560, 640, False, None),
              ("cross kv C1280", 128, 2560, 2048, False, None), ("proj_in 1x1", 1024, 1280, 1280, False, None),
              ("conv C1280 32x32", 1024, 1280, 9 * 1280, False, ops.ConvGeom(1, 32, 32, 1280, 32, 32)),
              ("conv C640 64x64", 4096, 640, 9 * 640, False, ops.ConvGeom(1, 64, 64, 640, 64, 64)),
              ("conv C320 128x128", 16384, 320, 9 * 320, False, ops.ConvGeom(1, 128, 128, 320, 128, 128)),
              ("conv up 2560->1280", 1024, 1280, 9 * 2560, False, ops.ConvGeom(1, 32, 32, 2560, 32, 32)),
              ("conv 1920->640 64x64", 4096, 640, 9 * 1920, False, ops.ConvGeom(1, 64, 64, 1920, 64, 64)),
              ("conv 960->320 128", 16384, 320, 9 * 960, False, ops.ConvGeom(1, 128, 128, 960, 128, 128)),
              ("clip qkv M=128", 128, 3840, 1280, False, None), ("clip o M=128", 128, 1280, 1280, False, None), ("clip fc1 M=128", 128, 5120, 1280, False, None), ("clip fc2", 128, 1280, 5120, False, None)]
    for (name, M, N, K, lora, conv) in shapes:
        res = []
        for tile in (1, 2, 3, 4, 6):
          for sk in (1, 2, 3, 4):
            if sk > 1 and (M * N > 1024 * 1280 * 2 or tile in (4, 6)): continue
            for st in (0, 2):
                try:
                    us = bench(M, N, K, tile, sk, lora, conv, stages=st)
                except Exception as e:
                    continue
                res.append((us, tile, st, sk))
        res.sort()
        auto = bench(M, N, K, 0, 0, lora, conv)
        fl = 2.0 * M * N * K
        print(f"{name:22s} M{M} N{N} K{K} lora{int(lora)}: auto {auto:.1f}us ({fl/auto/1e6:.0f} TF) | best " + ", ".join(f"t{t}/st{k}/sk{q}:{u:.1f}" for u, t, k, q in res[:5]) + f" | worst {res[-1][0]:.1f}")
