"""Scratch: per-iteration clock stamps of the attention forward (library built with -DSDLT_ATTN_TRACE): wave 0 of workgroup 0 writes
clock64() at the loop top, after the softmax, after the P.V MFMAs were issued, after the next tile's LDS stores, and after the loop."""
import sys, os, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sd_lora_trainer_amd import ops
BF = torch.bfloat16
for (name, B, H, Nq, Nk, d) in [("self N1024 H20", 1, 20, 1024, 1024, 64), ("self N4096 H10", 1, 10, 4096, 4096, 64)]:
    C = H * d
    r = lambda n: torch.randn(B * n, C, device="cuda").to(BF)
    Q, K, V = r(Nq), r(Nk), r(Nk)
    O = torch.zeros(B * Nq, C, dtype=BF, device="cuda"); L = torch.zeros(B * H * Nq, device="cuda")
    D = torch.zeros(4096, dtype=torch.int64, device="cuda")
    kw = dict(B=B, H=H, Nq=Nq, Nk=Nk, Nqp=Nq, Nkp=Nk, d=d, scale=1 / math.sqrt(d))
    for rep in range(3):
        D.zero_()
        pp = ops._attn_params(Q, K, V, causal=False, **kw)
        pp.O, pp.ldo, pp.L, pp.D = ops._p(O), ops._ld(O), ops._p(L), ops._p(D)
        ops._lib.check(ops._lib.load().sdlt_attn_fwd(ops.C.byref(pp), ops._stream()), "sdlt_attn_fwd")
        torch.cuda.synchronize()
    t = D.cpu().tolist()
    t = [x for x in t if x]
    print(name, "stamps", len(t))
    base = t[0]
    rel = [x - base for x in t]
    print("  first 24 stamps (cycles from start):", rel[:24])
    per = [rel[i + 4] - rel[i] for i in range(1, len(rel) - 5, 4)]
    print("  per-iteration (top->top):", per[:20])
    seg = [[rel[i + k + 1] - rel[i + k] for i in range(1, len(rel) - 5, 4)] for k in range(4)]
    for k, nm in enumerate(["top->softmax done", "softmax->PV issued", "PV issued->stores done", "stores->next top (barrier)"]):
        print(f"  {nm:28s}", seg[k][:16])

# cross-attention backward: stamps = start, operands resident, then per query tile: q-major done, barrier, key-major done, stores + barrier; end
for (name, B, H, Nq) in [("cross N1024 H20", 1, 20, 1024), ("cross N4096 H10", 1, 10, 4096)]:
    d, Nk, Nkp = 64, 77, 80
    C = H * d
    r = lambda n: torch.randn(B * n, C, device="cuda").to(BF)
    Q, K, V, dO = r(Nq), r(Nkp), r(Nkp), r(Nq)
    O = torch.zeros(B * Nq, C, dtype=BF, device="cuda"); L = torch.zeros(B * H * Nq, device="cuda")
    kw = dict(B=B, H=H, Nq=Nq, Nk=Nk, Nqp=Nq, Nkp=Nkp, d=d, scale=1 / math.sqrt(d))
    ops.attn_fwd(Q, K, V, None, O, L, **kw)
    D = torch.zeros(4096, dtype=torch.int64, device="cuda")
    dQ, dK, dV = torch.zeros_like(Q), torch.zeros_like(K), torch.zeros_like(V)
    qs = max(2, min((Nq + 63) // 64, 160 // (H * B)))
    dK32, dV32 = torch.empty(qs * B * Nkp, C, device="cuda"), torch.empty(qs * B * Nkp, C, device="cuda")
    for rep in range(3):
        D.zero_()
        ops.attn_bwd(Q, K, V, None, None, O, L, dO, None, D.view(torch.float32), dQ, dK, dV, qsplit=qs, dK32=dK32, dV32=dV32, **kw)
        torch.cuda.synchronize()
    allst = D.cpu().tolist()
    for role, seg in (("q-major role / single", allst[:64]), ("key-major role", allst[64:128])):
        t = [x for x in seg if x]
        if t:
            print(name, "qsplit", qs, role, "stamps (clock64 ticks, ~1.5 per ns):", [x - t[0] for x in t])
