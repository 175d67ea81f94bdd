"""Scratch: phase clock stamps of the tiled GEMM (library built with -DSDLT_GEMM_TRACE): workgroup 0 / thread 0 stamps clock64() at kernel
start (0), index math done (1), ring prefill issued (2), the first four K steps visible (3..6), main loop done (8), split-K reduction done (9),
adapter up-projection done (10), epilogue done (11)."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sd_lora_trainer_amd import ops, _lib
BF = torch.bfloat16
lib = _lib.load()
os.environ.setdefault("SDLT_WSK", "0")
ops.WSK = False
ops.WSK_LORA = False
def run(name, M, N, K, lora=False, res=False, geglu=False, **kw):
    X = torch.randn(M, K, device="cuda").to(BF); W = (torch.randn(N, K, device="cuda") * 0.02).to(BF)
    Y = torch.empty(M, N // 2 if geglu else N, device="cuda", dtype=BF); R = torch.randn(M, N, device="cuda").to(BF) if res else None
    if lora:
        kw["lora"] = ((torch.randn(16, K, device="cuda") / 16).to(BF), (torch.randn(N, 16, device="cuda") * 0.01).to(BF), 1.0, None)
    for rep in range(4):
        ops.gemm(X, W, Y, residual=R, **kw)
        torch.cuda.synchronize()
    out = (ctypes.c_longlong * 16)()
    lib.sdlt_gemm_trace_read(out)
    t = list(out)
    rel = [x - t[0] for x in t[:12]]
    r2 = [t[12] - t[0], t[13] - t[0], t[14] - t[0]]
    print(f"{name:34s} tile-coords {r2[0]:5d} geometry {r2[1]:5d} epi-prefetch {r2[2]:5d} index {rel[1]:6d} prefill {rel[2]:6d} steps {rel[3:7]} loop end {rel[8]:7d} splitk {rel[9]:7d} lora-up {rel[10]:7d} end {rel[11]:7d}")
def run_ff(name, M, N, K, mode):
    """the feed-forward products with their fused GEGLU epilogues (round 6: the three largest gaps of tools/step_floor.py are the FF chain)"""
    X = torch.randn(M, K, device="cuda").to(BF); W = (torch.randn(N, K, device="cuda") * 0.02).to(BF)
    bias = torch.randn(N, device="cuda")
    kw = {}
    if mode == "geglu":            # ff.net.0.proj: F1 [M, N] and hidden * gelu(gate) [M, N / 2]
        Y = torch.empty(M, N, device="cuda", dtype=BF)
        kw = dict(geglu_out=torch.empty(M, N // 2, device="cuda", dtype=BF), bias=bias)
    elif mode == "geglu_bwd":      # dX of ff.net.2 with GEGLU's backward in the epilogue: reads F1 [M, 2N], writes dF1 [M, 2N]
        Y = None
        kw = dict(geglu_bwd=(torch.randn(M, 2 * N, device="cuda").to(BF), torch.empty(M, 2 * N, device="cuda", dtype=BF)))
    else:
        Y = torch.empty(M, N, device="cuda", dtype=BF)
        kw = dict(bias=bias)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for rep in range(4):
        e0.record(); ops.gemm(X, W, Y, **kw); e1.record()
        torch.cuda.synchronize()
    out = (ctypes.c_longlong * 16)()
    lib.sdlt_gemm_trace_read(out)
    t = list(out)
    rel = [x - t[0] for x in t[:12]]
    print(f"{name:34s} eager launch {e0.elapsed_time(e1) * 1e3:6.1f} us | index {rel[1]:6d} prefill {rel[2]:6d} steps {rel[3:7]} loop end {rel[8]:7d} splitk {rel[9]:7d} lora-up {rel[10]:7d} end {rel[11]:7d} (ticks of workgroup 0)")
if os.environ.get("SDLT_TRACE_FF") == "1":
    run_ff("ff.net.0 1024x10240x1280 plain", 1024, 10240, 1280, "plain")
    run_ff("ff.net.0 1024x10240x1280 geglu", 1024, 10240, 1280, "geglu")
    run_ff("ff.net.2 dX 1024x5120x1280 plain", 1024, 5120, 1280, "plain")
    run_ff("ff.net.2 dX 1024x5120x1280 geglu-bwd", 1024, 5120, 1280, "geglu_bwd")
    run_ff("4096x5120x640 geglu", 4096, 5120, 640, "geglu")
    sys.exit(0)
run("1024x1280x1280 lora", 1024, 1280, 1280, lora=True)
run("1024x1280x1280 lora res", 1024, 1280, 1280, lora=True, res=True)
run("1024x3840x1280", 1024, 3840, 1280)
run("1024x1280x10240", 1024, 1280, 10240)
run("1024x1280x5120 res", 1024, 1280, 5120, res=True)
run("1024x10240x1280", 1024, 10240, 1280)
run("4096x640x640 lora", 4096, 640, 640, lora=True)
run("4096x640x2560 res", 4096, 640, 2560, res=True)
run("16384x320x2880", 16384, 320, 2880)
