"""Lab (round 6): does touching the NEXT wave-split-K product's packed weight from the current launch (sdlt_wsk_gemm_params.pf_next_w) shorten a chain of dependent launches?
A chain of n launches Y_i = X_i W_i^T (+ rank-16 adapter) whose weights rotate through > 256 MB (every launch streams a weight that is in no cache, like in the step) and whose
input is the previous launch's output (a dependent chain, like in the step); graph-replayed; with the hint off and with the first S K steps of the next weight touched."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sd_lora_trainer_amd import _lib, ops

BF = torch.bfloat16
lib = _lib.load()


def chain(M, N, K, lora, steps, n=48, reps=5):
    NROT = max(n, (320 << 20) // (N * K * 2))
    ws = [(torch.randn(N, K, device="cuda") * K ** -0.5).to(BF) for _ in range(NROT)]
    wps = []
    st = torch.cuda.current_stream().cuda_stream
    for w in ws:
        wp = torch.empty(N * K, dtype=BF, device="cuda")
        _lib.check(lib.sdlt_wsk_pack_weight(w.data_ptr(), K, N, K, wp.data_ptr(), st), "pack")
        wps.append(wp)
    del ws
    A, Bu = (torch.randn(16, K, device="cuda") / 16).to(BF), (torch.randn(N, 16, device="cuda") * 0.05).to(BF)
    xs = [torch.randn(M, K, device="cuda").to(BF) for _ in range(2)] if K != N else None
    bufs = [torch.randn(M, N, device="cuda").to(BF) for _ in range(2)]
    T = torch.zeros(M, 16, dtype=BF, device="cuda")

    def launch(i, hint):
        q = _lib.WskGemmParams()
        x = bufs[i & 1] if K == N else xs[i & 1]
        q.X, q.ldx, q.W, q.ldw, q.M, q.N, q.K = x.data_ptr(), K, wps[i % NROT].data_ptr(), 0, M, N, K
        q.Y, q.ldy = bufs[(i + 1) & 1].data_ptr(), N
        if lora:
            q.Adown, q.ld_adown, q.Bup, q.ld_bup, q.lora_scale, q.lora_rp, q.T_out, q.ld_t = A.data_ptr(), K, Bu.data_ptr(), 16, 0.01, 16, T.data_ptr(), 16
        if hint:
            q.pf_next_w, q.pf_next_n, q.pf_next_k, q.pf_steps = wps[(i + 1) % NROT].data_ptr(), N, K, steps
        _lib.check(lib.sdlt_wsk_gemm_p(C.byref(q), torch.cuda.current_stream().cuda_stream), "wsk_gemm_p")

    out = []
    for hint in (False, True):
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            for i in range(n):
                launch(i, hint)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for i in range(n):
                launch(i, hint)
        best = 1e9
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / n * 1e3)
        out.append(best)
    return out


for (M, N, K, lora) in [(1024, 1280, 1280, True), (1024, 1280, 1280, False), (1024, 1280, 5120, False), (1024, 1280, 10240, False), (1024, 1280, 3840, True)]:
    line = f"M{M} N{N} K{K} {'lora16' if lora else '      '}:"
    for S in (4, 8, 20, 40):
        if S > K // 64 and S != 20:
            continue
        off, on = chain(M, N, K, lora, S)
        line += f"  S={S}: {off:6.2f} -> {on:6.2f} us ({100 * (on / off - 1):+5.1f} %)"
    print(line, flush=True)
