"""K-grouped (fused dX) GEMM timing per tile for the SDXL shapes."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sd_lora_trainer_amd import ops
BF = torch.bfloat16
def timeit(fn, reps=20):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (5 * reps)
for (M, N, C, G) in [(4096, 640, 640, 3), (1024, 1280, 1280, 3), (128, 2048, 1280, 2), (128, 2048, 640, 2)]:
    K = G * C
    X = torch.randn(M, K, device="cuda").to(BF); Ws = [(torch.randn(N, K, device="cuda") * 0.02).to(BF) for _ in range(4)]
    Ad = torch.randn(16, K, device="cuda").to(BF); Bu = torch.randn(N, G * 16, device="cuda").to(BF)
    out = torch.empty(M, N, device="cuda", dtype=BF); T = torch.empty(M, G * 16, device="cuda", dtype=BF)
    res = []
    for tile in (0, 1, 2, 3):
        for sk in (0, 1):
            i = [0]
            def f():
                i[0] += 1
                ops.gemm(X, Ws[i[0] % 4], out, lora=(Ad, Bu, 1.0, T), lora_group_k=C, tile=tile, splitk=sk)
            try: res.append(f"t{tile}/sk{sk}:{timeit(f):.1f}")
            except Exception as e: res.append(f"t{tile}/sk{sk}:ERR")
    print(M, N, K, " ".join(res))
