"""Is hipGraphLaunch asynchronous for a big graph?  Host time per replay() vs GPU time per replay."""
import torch, time
x = torch.randn(512, 512, device="cuda")
def body(n):
    y = x
    for _ in range(n):
        y = torch.tanh(y)
    return y
for n in (200, 3000):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s): body(3)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g): body(n)
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2): body(n)
    g.replay(); g2.replay(); torch.cuda.synchronize()
    for name, seq in (("same exec", [g] * 6), ("alternating", [g, g2] * 3)):
        torch.cuda.synchronize()
        t0 = time.perf_counter(); host = []
        for gg in seq:
            t = time.perf_counter(); gg.replay(); host.append((time.perf_counter() - t) * 1e3)
        torch.cuda.synchronize()
        tot = (time.perf_counter() - t0) * 1e3
        print(f"{n} nodes, {name}: host ms per replay {[round(h, 2) for h in host]}, total {tot:.2f} ms for {len(seq)} replays")
