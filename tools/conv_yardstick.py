"""Yardstick, NOT a product path: the vendor library's 3x3 convolution (torch conv2d, channels_last bf16 -> MIOpen) on the UNet's dominant conv
shapes, weights rotating through HBM, against this repo's implicit-GEMM kernel on the same operands.  python tools/conv_yardstick.py"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sd_lora_trainer_amd import ops as O

BF = torch.bfloat16
torch.backends.cudnn.benchmark = True
shapes = [(1, 32, 32, 1280, 1280), (1, 32, 32, 2560, 1280), (1, 64, 64, 640, 640), (1, 64, 64, 1280, 640), (1, 128, 128, 320, 320), (1, 128, 128, 640, 320), (1, 64, 64, 1920, 640)]
for (B, H, W, Ci, Co) in shapes:
    nW = max(4, min(40, int(600e6 / (Co * Ci * 9 * 2))))
    ws = [torch.randn(Co, Ci, 3, 3, device="cuda", dtype=BF) * (9 * Ci) ** -0.5 for _ in range(nW)]
    wcl = [w.contiguous(memory_format=torch.channels_last) for w in ws]
    wm = [w.permute(0, 2, 3, 1).reshape(Co, 9 * Ci).contiguous() for w in ws]          # [N, tap*Cin + ci]
    x = torch.randn(B, Ci, H, W, device="cuda", dtype=BF).contiguous(memory_format=torch.channels_last)
    xn = x.permute(0, 2, 3, 1).reshape(B * H * W, Ci).contiguous()
    out = torch.empty(B * H * W, Co, device="cuda", dtype=BF)
    geom = O.ConvGeom(B, H, W, Ci, H, W)
    res = {}
    for name, fn in (("miopen", lambda i: torch.nn.functional.conv2d(x, wcl[i], padding=1)), ("sdlt", lambda i: O.gemm(xn, wm[i], out, conv=geom))):
        for i in range(min(3, nW)):
            fn(i)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for i in range(nW):
                fn(i)
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            g.replay()
        e1.record(); torch.cuda.synchronize()
        res[name] = e0.elapsed_time(e1) * 1e3 / (5 * nW)
    fl = 2.0 * B * H * W * Co * Ci * 9
    ref = torch.nn.functional.conv2d(x, wcl[0], padding=1).permute(0, 2, 3, 1).reshape(B * H * W, Co).float()
    O.gemm(xn, wm[0], out, conv=geom)
    err = float((out.float() - ref).abs().max() / ref.abs().max())
    print(f"{H}x{W} {Ci}->{Co}: vendor {res['miopen']:6.1f} us ({fl / res['miopen'] / 1e6:5.0f} TF/s)   this repo {res['sdlt']:6.1f} us ({fl / res['sdlt'] / 1e6:5.0f} TF/s)   x{res['miopen'] / res['sdlt']:.2f}   (max rel diff {err:.3g})")
