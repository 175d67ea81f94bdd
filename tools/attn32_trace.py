"""Scratch: clock stamps inside the 32-rows-per-wave attention kernels (library built with -DSDLT_ATTN32_TRACE, tools/lab_build_attn32.sh
trace:-DSDLT_ATTN32_TRACE; run with SDLT_KERNEL_LIB=tools/lab/lib_trace.so).  Lane 0 of wave 0 of the first workgroup of a role writes clock64()
  forward : start | per iteration: top, DMA issued, S MFMAs issued, softmax done, (P V issued) end of body | after the loop
  backward: start | per iteration: top, DMA issued, then per 32-row block: first products issued, softmax arithmetic done, block done | after the loop
The kernel's duration (events) calibrates the tick."""
import sys, os, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sd_lora_trainer_amd import ops
BF = torch.bfloat16

def show(name, t, per_it, labels):
    t = [x for x in t if x]
    if len(t) < 3:
        print(name, "no stamps"); return
    rel = [x - t[0] for x in t]
    n_it = (len(rel) - 2) // per_it
    print(f"{name}: {len(t)} stamps, {n_it} iterations, first stamp -> loop start {rel[1]}, total {rel[-1]} ticks")
    segs = [[rel[1 + i * per_it + k + 1] - rel[1 + i * per_it + k] for i in range(n_it) if 1 + i * per_it + k + 1 < len(rel)] for k in range(per_it)]
    for k in range(per_it):
        v = segs[k]
        print(f"   {labels[k]:34s} median {sorted(v)[len(v) // 2]:6d}   first 10: {v[:10]}")
    tops = [rel[1 + i * per_it] for i in range(n_it)]
    d = [b - a for a, b in zip(tops, tops[1:])]
    if d:
        print(f"   top -> top                         median {sorted(d)[len(d) // 2]:6d}   first 10: {d[:10]}")

for (name, B, H, N) in [("self N1024 H20", 1, 20, 1024), ("self N4096 H10", 1, 10, 4096)]:
    d = 64; C = H * d
    r = lambda n: torch.randn(B * n, C, device="cuda").to(BF)
    Q, K, V, dO = r(N), r(N), r(N), r(N)
    O = torch.zeros(B * N, C, dtype=BF, device="cuda"); L = torch.zeros(B * H * N, device="cuda")
    kw = dict(B=B, H=H, Nq=N, Nk=N, Nqp=N, Nkp=N, d=d, scale=1 / math.sqrt(d))
    st = torch.zeros(4096, dtype=torch.int64, device="cuda")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for rep in range(3):
        st.zero_()
        pp = ops._attn_params(Q, K, V, causal=False, **kw)
        pp.O, pp.ldo, pp.L, pp.D = ops._p(O), ops._ld(O), ops._p(L), ops._p(st)
        e0.record()
        ops._lib.check(ops._lib.load().sdlt_attn_fwd(ops.C.byref(pp), ops._stream()), "sdlt_attn_fwd")
        e1.record(); torch.cuda.synchronize()
    print(f"== {name} forward: {e0.elapsed_time(e1) * 1e3:.1f} us (one eager launch)")
    show("forward", st.cpu().tolist()[:512], 5, ["top -> DMA issued", "DMA issued -> S issued", "S issued -> softmax done", "softmax -> body end (PV issued)", "body end -> next top (barrier)"])
    ops.attn_fwd(Q, K, V, None, O, L, **kw)
    D = torch.zeros(B * H * N, device="cuda")
    dQ, dK, dV = torch.zeros_like(Q), torch.zeros_like(K), torch.zeros_like(V)
    for rep in range(3):
        st.zero_()
        e0.record()
        pp = ops._attn_params(Q, K, V, causal=False, **kw)
        pp.O, pp.ldo, pp.L, pp.dO, pp.lddo, pp.D = ops._p(O), ops._ld(O), ops._p(L), ops._p(dO), ops._ld(dO), ops._p(D)
        pp.dQ, pp.lddq, pp.dK, pp.lddk, pp.dV, pp.lddv = ops._p(dQ), ops._ld(dQ), ops._p(dK), ops._ld(dK), ops._p(dV), ops._ld(dV)
        pp.dK32 = ops._p(st)         # (unused by the self-attention kernels: the trace build stamps into it)
        ops._lib.check(ops._lib.load().sdlt_attn_bwd(ops.C.byref(pp), ops._stream()), "sdlt_attn_bwd")
        e1.record(); torch.cuda.synchronize()
    print(f"== {name} backward (D pre-pass + both roles): {e0.elapsed_time(e1) * 1e3:.1f} us")
    allst = st.cpu().tolist()
    lab = ["top -> DMA issued", "DMA issued -> blk0 products issued", "blk0 softmax arithmetic", "blk0 second products issued", "blk0 end -> blk1 products issued", "blk1 softmax arithmetic",
           "blk1 second products issued", "blk1 end -> next top (barrier)"]
    show("dQ role", allst[:512], 8, lab)
    show("dK/dV role", allst[512:1024], 8, lab)
