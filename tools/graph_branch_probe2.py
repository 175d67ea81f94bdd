"""Same question as graph_branch_probe.py, with this library's kernels: chains of M=128 GEMMs (+ layernorm) on two streams."""
import sys, os, torch, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sd_lora_trainer_amd import ops
dev = "cuda"
BF = torch.bfloat16
def mk(D): return dict(x=torch.randn(128, D, device=dev).to(BF), y=torch.empty(128, D, device=dev, dtype=BF), W=(torch.randn(D, D, device=dev) * 0.02).to(BF))
A, B = mk(1280), mk(768)
def chain(t, n=150, kind="gemm"):
    for _ in range(n):
        if kind == "gemm":
            ops.gemm(t["x"], t["W"], t["y"]); ops.gemm(t["y"], t["W"], t["x"])
        else:
            torch.tanh_(t["x"]); torch.tanh_(t["y"])
sides = [torch.cuda.Stream(), torch.cuda.Stream()]
def body(mode, kind):
    cur = torch.cuda.current_stream()
    for _ in range(PREFIX): torch.tanh_(A["y"])
    if mode == "serial":
        chain(A, kind=kind); chain(B, kind=kind)
    else:
        for s in sides: s.wait_stream(cur)
        for s, t in zip(sides, (A, B)):
            with torch.cuda.stream(s): chain(t, kind=kind)
        for s in sides: cur.wait_stream(s)
    for _ in range(SUFFIX): torch.tanh_(A["y"])
SUFFIX = 1
for PREFIX, SUFFIX in ((1, 3000), (300, 1), (1000, 1)):
 for kind in ("gemm",):
    for mode in ("serial", "side+side"):
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            body(mode, kind)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            body(mode, kind)
        g.replay(); torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(10): g.replay()
        torch.cuda.synchronize()
        print(f"prefix {PREFIX} suffix {SUFFIX} {kind:8s} {mode:10s} {(time.perf_counter() - t) / 10 * 1e3:.3f} ms per replay")
