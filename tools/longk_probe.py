import sys, os, torch
sys.path.insert(0, "/root/repo")
from sd_lora_trainer_amd import ops as O
BF = torch.bfloat16
def timeit(fn, Ws):
    for w in Ws[:3]: fn(w)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for w in Ws: fn(w)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (5 * len(Ws))
for (M, N, K) in [(1024, 1280, 10240), (1024, 1280, 5120)]:
    nW = 24
    Ws = [torch.randn(N, K, device="cuda", dtype=BF) * K ** -0.5 for _ in range(nW)]
    x = torch.randn(M, K, device="cuda", dtype=BF); r = torch.randn(M, N, device="cuda", dtype=BF)
    out = torch.empty(M, N, device="cuda", dtype=BF)
    print(M, N, K, "default(wsk)", round(timeit(lambda w: O.gemm(x, w, out, residual=r), Ws), 1))
    for tile, sk in [(8, 4), (8, 2), (8, 8), (7, 4), (7, 8), (1, 3), (1, 4), (4, 4), (4, 8), (6, 8), (6, 12)]:
        try:
            t = timeit(lambda w: O.gemm(x, w, out, residual=r, tile=tile, splitk=sk), Ws)
            print("   tile", tile, "splitk", sk, round(t, 1))
        except Exception as e:
            print("   tile", tile, "splitk", sk, "failed", str(e)[:80])
