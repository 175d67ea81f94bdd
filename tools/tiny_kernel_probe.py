"""Per-kernel cost of dependent tiny kernels inside a hipGraph: this library's kernels vs torch elementwise kernels."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sd_lora_trainer_amd import ops
dev = "cuda"; BF = torch.bfloat16
x = torch.randn(1024, 1280, device=dev).to(BF); y = torch.empty_like(x); st = torch.empty(1024 * 2, device=dev)
g = torch.ones(1280, device=dev); b = torch.zeros(1280, device=dev)
small = torch.randn(128, 1280, device=dev).to(BF); small2 = torch.empty_like(small)
z = torch.empty(4096, device=dev)
cases = {
    "torch tanh_ 128x1280": lambda: torch.tanh_(small),
    "torch zero_ 4096 f32": lambda: z.zero_(),
    "map_bf16 gelu 128x1280": lambda: ops.map_bf16(ops.MAP_GELU, small, None, small2),
    "add2d 128x1280": lambda: ops.add2d(small, small2, small2),
    "layernorm_fwd 1024x1280": lambda: ops.layernorm_fwd(x, y, st, gamma=g, beta=b),
    "torch tanh_ 1024x1280": lambda: torch.tanh_(x),
    "gemm 128x1280x1280": lambda: ops.gemm(small, W, small2),
}
W = (torch.randn(1280, 1280, device=dev) * 0.02).to(BF)
for name, fn in cases.items():
    N = 500
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3): fn()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(N): fn()
    gr.replay(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(5): gr.replay()
    torch.cuda.synchronize()
    print(f"{name:28s} {(time.perf_counter() - t) / 5 / N * 1e6:6.2f} us per kernel")
