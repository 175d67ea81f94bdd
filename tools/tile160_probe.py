"""160-column 8-wave tiles (7 = 256x160, 8 = 128x160) on the SDXL shapes: correctness against the default tile choice, then time
per launch (hipGraph hot loop, rotating weights) for auto vs 7 / 8 with split-K options."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gemm_probe import bench
from sd_lora_trainer_amd import ops
BF = torch.bfloat16

def check():
    torch.manual_seed(0)
    bad = 0
    for (M, N, K, lora, res, conv) in [(1024, 1280, 1280, True, True, None), (1024, 10240, 1280, False, False, None), (4096, 640, 640, True, False, None),
                                       (1000, 320, 320, False, True, None), (1024, 3840, 1280, "gn", False, None), (4096, 640, 5760, True, True, ops.ConvGeom(1, 64, 64, 640, 64, 64)),
                                       (16384, 320, 2880, False, False, ops.ConvGeom(1, 128, 128, 320, 128, 128))]:
        X = (torch.randn(M if conv is None else conv.B * conv.Hin * conv.Win, K if conv is None else conv.Cin, device="cuda") * 0.5).to(BF)
        W = (torch.randn(N, K, device="cuda") / K ** 0.5).to(BF)
        bias = torch.randn(N, device="cuda")
        R = torch.randn(M, N, device="cuda").to(BF) if res else None
        lo, kw = None, {}
        if lora == "gn":
            G = 3
            lo = ((torch.randn(G * 16, K, device="cuda") / 16).to(BF), (torch.randn(N, 16, device="cuda") * 0.1).to(BF), 1.0, torch.empty(M, G * 16, device="cuda", dtype=BF))
            kw = dict(lora_group_n=N // G)
        elif lora:
            lo = ((torch.randn(16, K, device="cuda") / 16).to(BF), (torch.randn(N, 16, device="cuda") * 0.1).to(BF), 1.0, torch.empty(M, 16, device="cuda", dtype=BF))
        ref = torch.empty(M, N, device="cuda", dtype=BF)
        ops.gemm(X, W, ref, lora=lo, bias=bias, residual=R, conv=conv, **kw)
        Tref = lo[3].clone() if lo else None
        for tile in (7, 8):
            for sk in (1, 2):
                if sk > 1 and (K // 64) < 16: continue
                out = torch.empty(M, N, device="cuda", dtype=BF)
                if lo: lo[3].zero_()
                ops.gemm(X, W, out, lora=lo, bias=bias, residual=R, conv=conv, tile=tile, splitk=sk, **kw)
                torch.cuda.synchronize()
                err = float((out.float() - ref.float()).abs().max()) / float(ref.float().abs().max())
                terr = float((lo[3].float() - Tref.float()).abs().max()) if lo else 0.0
                ok = err < 1e-2 and terr < 2e-2
                bad += not ok
                print(f"check M{M} N{N} K{K} lora={lora} res={res} conv={conv is not None} tile{tile} sk{sk}: rel err {err:.2e} T err {terr:.2e} {'ok' if ok else 'FAIL'}")
    return bad

if __name__ == "__main__":
    bad = check()
    print("FAILURES:", bad)
    shapes = [("ff1 C1280", 1024, 10240, 1280, False, None), ("ff2 bwd C1280", 1024, 5120, 1280, False, None), ("qkv C1280", 1024, 3840, 1280, False, None),
              ("attn proj C1280 lora", 1024, 1280, 1280, True, None), ("ff2 C1280", 1024, 1280, 5120, False, None), ("ff1 bwd C1280", 1024, 1280, 10240, False, None),
              ("ff1 C640", 4096, 5120, 640, False, None), ("ff2 bwd C640", 4096, 2560, 640, False, None), ("qkv C640", 4096, 1920, 640, False, None),
              ("attn proj C640 lora", 4096, 640, 640, True, None), ("ff2 C640", 4096, 640, 2560, False, None), ("ff1 bwd C640", 4096, 640, 5120, False, None),
              ("conv C320 128x128", 16384, 320, 2880, False, ops.ConvGeom(1, 128, 128, 320, 128, 128)), ("conv C640 64x64", 4096, 640, 5760, False, ops.ConvGeom(1, 64, 64, 640, 64, 64)),
              ("conv C1280 32x32", 1024, 1280, 11520, False, ops.ConvGeom(1, 32, 32, 1280, 32, 32)), ("conv 960->320 128", 16384, 320, 8640, False, ops.ConvGeom(1, 128, 128, 960, 128, 128)),
              ("conv 1920->640", 4096, 640, 17280, False, ops.ConvGeom(1, 64, 64, 1920, 64, 64)), ("conv 2560->1280", 1024, 1280, 23040, False, ops.ConvGeom(1, 32, 32, 2560, 32, 32)),
              ("sd15 ff1 C320 b4", 16384, 2560, 320, False, None), ("sd15 proj C320 b4 lora", 16384, 320, 320, True, None), ("sd15 ff2 C320 b4", 16384, 320, 1280, False, None)]
    for (name, M, N, K, lora, conv) in shapes:
        auto = bench(M, N, K, 0, 0, lora, conv)
        res = []
        for tile in (7, 8):
            for sk in (1, 2, 3, 4, 6):
                if sk > 1 and (K // 64) // sk < 6: continue
                for st in (0, 2):
                    try: us = bench(M, N, K, tile, sk, lora, conv, stages=st)
                    except Exception as e: continue
                    res.append((us, tile, sk, st))
        res.sort()
        fl = 2.0 * M * N * K
        print(f"{name:24s} M{M} N{N} K{K}: auto {auto:.1f}us ({fl / auto / 1e6:.0f} TF) | " + ", ".join(f"t{t}/sk{q}/st{k}:{u:.1f}" for u, t, q, k in res[:6]) + f" | best {fl / res[0][0] / 1e6:.0f} TF")
