"""Does a captured fork/join run its branches concurrently on this ROCm?  Independent chains of small kernels, forked
at the graph root or after a common prefix, with the branches on (cur + side) or on (side + side)."""
import torch, time
dev = "cuda"
a = [torch.randn(256, 256, device=dev) for _ in range(3)]
def chain(x, n=200):
    for _ in range(n):
        x = torch.tanh(x @ x * 0.01)
    return x
sides = [torch.cuda.Stream(), torch.cuda.Stream()]
def body(mode, prefix):
    cur = torch.cuda.current_stream()
    x0 = chain(a[2], 20) if prefix else None
    if mode == "serial":
        chain(a[0]); chain(a[1])
    elif mode == "cur+side":
        sides[0].wait_stream(cur)
        with torch.cuda.stream(sides[0]): chain(a[1])
        chain(a[0])
        cur.wait_stream(sides[0])
    else:
        for s in sides: s.wait_stream(cur)
        for s, t in zip(sides, a):
            with torch.cuda.stream(s): chain(t)
        for s in sides: cur.wait_stream(s)
    if prefix: chain(a[2], 20)
for prefix in (False, True):
    for mode in ("serial", "cur+side", "side+side"):
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            body(mode, prefix)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            body(mode, prefix)
        g.replay(); torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(10): g.replay()
        torch.cuda.synchronize()
        print(f"prefix={prefix} {mode:10s} {(time.perf_counter() - t) / 10 * 1e3:.3f} ms per replay")
