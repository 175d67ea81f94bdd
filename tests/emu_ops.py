"""TEST-ONLY emulation of the C-ABI ops (sd-lora-trainer_amd/ops.py) in plain torch on CPU.

Two uses, both as a CHECKER and never as a product path:
  * `-m "not gpu"` tests drive the host-side plan (sd-lora-trainer_amd/unet.py, step.py) through this table
    to validate its explicit backward against the oracle's autograd;
  * `-m gpu` tests use the same functions as the per-kernel reference for the HIP kernels.
Each function documents the op's contract by restating it; math is done in fp32 and rounded to the
dtype of the output buffer.
"""
import math

import torch
import torch.nn.functional as F

F32 = torch.float32
MAP_SILU, MAP_DSILU, MAP_ADD, MAP_GELU, MAP_DGELU, MAP_QGELU, MAP_DQGELU = range(7)


class ConvGeom:
    __slots__ = ("B", "Hin", "Win", "Cin", "Hout", "Wout", "stride", "ups", "flip", "tr")

    def __init__(self, B, Hin, Win, Cin, Hout, Wout, stride=1, ups=1, flip=0, tr=0):
        self.B, self.Hin, self.Win, self.Cin, self.Hout, self.Wout = B, Hin, Win, Cin, Hout, Wout
        self.stride, self.ups, self.flip, self.tr = stride, ups, flip, tr


def _conv_apply(X, Wm, g):
    """X [B*Hin*Win, Cin] NHWC, Wm [N, 9*Cin] with k = tap*Cin + ci -> [B*Hout*Wout, N] (fp32)."""
    N = Wm.shape[0]
    x = X.float().reshape(g.B, g.Hin, g.Win, g.Cin).permute(0, 3, 1, 2)
    wk = Wm.float().reshape(N, 3, 3, g.Cin).permute(0, 3, 1, 2)   # [N, Cin, dy, dx]
    if g.tr:
        assert g.flip
        y = F.conv_transpose2d(x, wk.permute(1, 0, 2, 3), stride=2, padding=1, output_padding=1)
    else:
        if g.ups == 2:
            x = F.interpolate(x, scale_factor=2.0, mode="nearest")
        if g.flip:
            wk = torch.flip(wk, dims=(2, 3))
        y = F.conv2d(x, wk, stride=g.stride, padding=1)
    assert y.shape[2] == g.Hout and y.shape[3] == g.Wout, (y.shape, g.Hout, g.Wout)
    return y.permute(0, 2, 3, 1).reshape(g.B * g.Hout * g.Wout, N)


def set_throughput_hint(flag):
    pass


def gemm(X, W, out, *, X2=None, W2=None, conv=None, lora=None, bias=None, rowbias=None, rows_per_batch=0,
         residual=None, alpha=1.0, Ct=None, tile=0, splitk=0, stages=0, accumulate=False, lora_group_n=0, lora_group_k=0, batch=None,
         geglu_out=None, geglu_bwd=None, act_out=None, dact_in=None, col_scale=None, ln=None, ln_parts_out=None, rowdot=None, out0=None):
    if out0 is not None:        # sdlt_wsk_gemm_params.Y0: the layer's own output (rounded) before the residual, and out = its unrounded value + residual
        gemm(X, W, out0, lora=lora, bias=bias, col_scale=col_scale, lora_group_n=lora_group_n, lora_group_k=lora_group_k)
        gemm(X, W, out, lora=lora, bias=bias, col_scale=col_scale, residual=residual, lora_group_n=lora_group_n, lora_group_k=lora_group_k, ln_parts_out=ln_parts_out)
        return out
    if rowdot is not None:      # sdlt_wsk_gemm_rowdot: the product, then D[b, h, q] += sum over head h's columns of rounded(out) o O
        assert residual is None and conv is None and ln is None and ln_parts_out is None and not accumulate
        gemm(X, W, out, lora=lora, bias=bias, alpha=alpha, Ct=Ct, lora_group_k=lora_group_k)
        M, N = out.shape
        O_, D_, Nq = rowdot["O"], rowdot["D"], rowdot["Nq"]
        H = D_.numel() // M
        prod = (out.float() * O_.float()).reshape(M // Nq, Nq, H, N // H).sum(-1)           # [B, Nq, H]
        D_.view(M // Nq, H, Nq).add_(prod.permute(0, 2, 1))
        rowdot["done"] = True
        return out
    if ln is not None:      # folded LayerNorm (sdlt_gemm_params.ln_c1): raw rows in, W = W o gamma, bias = c2
        c1, stats_, eps_, lnad = ln[:4]
        assert conv is None and X2 is None and alpha == 1.0 and col_scale is None and not lora_group_k and batch is None
        xf = X.float()
        mean = xf.mean(1, keepdim=True)
        rstd = torch.rsqrt(xf.var(1, unbiased=False, keepdim=True) + eps_)
        if len(ln) > 4 and ln[4] is not None:      # statistics from the producer's row partials [M, P, 2]
            pp = ln[4].view(-1)[: X.shape[0] * ln[5] * 2].view(X.shape[0], ln[5], 2).float()      # (sum, centred sum of squares) per tile
            nt = X.shape[1] / ln[5]
            mean = pp[:, :, 0].sum(1, keepdim=True) / X.shape[1]
            m2 = pp[:, :, 1].sum(1, keepdim=True) + (nt * (pp[:, :, 0] / nt - mean) ** 2).sum(1, keepdim=True)
            rstd = torch.rsqrt(m2 / X.shape[1] + eps_)
        if stats_ is not None:
            stats_.view(-1)[: 2 * X.shape[0]].copy_(torch.cat([mean, rstd], 1).reshape(-1))
    if geglu_bwd is not None:     # dX of ff.net.2 fused with GEGLU's backward; F1 / dF1 in the interleaved-16 layout
        f1, df1 = geglu_bwd
        H = W.shape[0]
        perm = geglu_perm(H)
        dG = (X.float() @ W.float().t()) * alpha
        halves = torch.empty_like(f1, dtype=F32)
        halves[:, perm] = f1.float()                   # interleaved position p holds original row perm[p]
        x = halves.clone().requires_grad_(True)
        h, g_ = x.chunk(2, dim=1)
        (gr,) = torch.autograd.grad(h * F.gelu(g_), x, dG)
        df1.copy_(gr[:, perm].to(df1.dtype))
        return df1
    if batch is not None:     # batched launch: every problem with its own operands, same options
        for it in batch.items:
            lo = None if lora is None else (it.get("Adown", lora[0]) if it.get("Adown") is not None else lora[0], it.get("Bup") if it.get("Bup") is not None else lora[1], lora[2], it.get("T_out") if it.get("T_out") is not None else lora[3])
            gemm(it["X"] if it.get("X") is not None else X, it["W"] if it.get("W") is not None else W, it["C"] if it.get("C") is not None else out, X2=X2, W2=W2, conv=conv, lora=lo,
                 bias=it.get("bias") if it.get("bias") is not None else bias, rowbias=rowbias, rows_per_batch=rows_per_batch, residual=residual, alpha=alpha,
                 Ct=it.get("Ct") if it.get("Ct") is not None else Ct, accumulate=accumulate, lora_group_n=lora_group_n, lora_group_k=lora_group_k,
                 col_scale=it.get("col_scale") if it.get("col_scale") is not None else col_scale)
        return out
    acc = X.float() @ W.float().t() if conv is None else _conv_apply(X, W, conv)
    if ln is not None:
        acc = rstd * (acc - mean * c1.float()[None, :])
    if X2 is not None:
        acc = acc + X2.float() @ W2.float().t()
    if lora is not None:
        Adown, Bup, scale, T_out = lora
        if lora_group_k:   # stacked gradients: T_g = X_g . Adown[:, group g]^T, one 16-column block per group
            Gk = X.shape[1] // lora_group_k
            T = torch.cat([X[:, gi * lora_group_k:(gi + 1) * lora_group_k].float() @ Adown[:, gi * lora_group_k:(gi + 1) * lora_group_k].float().t()
                           for gi in range(Gk)], 1)
        else:
            T = X.float() @ Adown.float().t() if conv is None else _conv_apply(X, Adown, conv)
        if ln is not None:      # T = rstd (x (A o gamma)^T - mean cA) + A beta, constants [G][cA(16) | abeta(16)]
            cc = lnad.float().view(-1, 2, 16)
            T = rstd * (T - mean * cc[:, 0].reshape(1, -1)) + cc[:, 1].reshape(1, -1)
        Tb = (T * scale).to(out.dtype if out.dtype != F32 else X.dtype)
        if T_out is not None:
            T_out.copy_(Tb)
        if lora_group_n:   # stacked projections, one adapter per group of lora_group_n output columns
            Rp = Bup.shape[1]
            for gi in range(W.shape[0] // lora_group_n):
                cols = slice(gi * lora_group_n, (gi + 1) * lora_group_n)
                acc[:, cols] = acc[:, cols] + Tb[:, gi * Rp:(gi + 1) * Rp].float() @ Bup[cols].float().t()
        else:
            acc = acc + Tb.float() @ Bup.float().t()
    acc = acc * alpha
    if col_scale is not None:     # DoRA: magnitude / norm per output column, before bias / residual
        assert lora is not None
        acc = acc * col_scale.float()
    if bias is not None:
        acc = acc + bias.float()
    if rowbias is not None:
        idx = torch.arange(acc.shape[0]) // rows_per_batch
        acc = acc + rowbias.float()[idx]
    if residual is not None:
        acc = acc + residual.float()
    if accumulate:
        acc = acc + out.float()
    if dact_in is not None:       # out = product * act'(pre-activation)
        kind, pre = dact_in
        x = pre.detach().float().clone().requires_grad_(True)
        y = F.gelu(x) if kind == "gelu" else x * torch.sigmoid(1.702 * x)
        (dx,) = torch.autograd.grad(y, x, acc)
        acc = dx
    out.copy_(acc.to(out.dtype))
    if ln_parts_out is not None:      # (sum, centred sum of squares) of the rounded output row per 80-column tile
        o = out.float().view(out.shape[0], -1, _part_width(out.shape[1]))
        ln_parts_out.view(-1)[: o.shape[0] * o.shape[1] * 2].copy_(torch.stack([o.sum(2), ((o - o.mean(2, keepdim=True)) ** 2).sum(2)], 2).reshape(-1))
    if act_out is not None:
        kind, a_ = act_out
        a_.copy_((F.gelu(acc) if kind == "gelu" else acc * torch.sigmoid(1.702 * acc)).to(a_.dtype))
    if Ct is not None:
        Ct[:, : acc.shape[0]].copy_(acc.t().to(Ct.dtype))
    if geglu_out is not None:     # ff.net.0.proj in the interleaved-16 layout: hidden * gelu(gate)
        H = acc.shape[1] // 2
        halves = torch.empty_like(acc)
        halves[:, geglu_perm(H)] = acc
        h, g_ = halves.chunk(2, dim=1)
        geglu_out.copy_((h * F.gelu(g_)).to(geglu_out.dtype))
    return out


STRIP_ANY_SHAPE = True      # the emulation has no K % 256 rule: the CPU tests run the fused text-encoder plan on the toy widths too


def strip_partial_splits(N, K, B):
    return 2 if K >= 128 else 1


def strip_gemm(X, W, out, *, B, T, Tp, bias=None, residual=None, act_out=None, dact_in=None, ln=None, stats=None, splitk=None, partial=None):
    """sdlt_strip_gemm: only the t < T rows of every batch element are produced; with ln the product runs on the raw rows against
    (weight o gamma) and the LayerNorm is applied algebraically from the row statistics."""
    rows = (torch.arange(B)[:, None] * Tp + torch.arange(T)[None, :]).reshape(-1)
    x = X.float()[rows]
    if partial is not None:      # the K split's tiles as they are: slab s holds the product over its share of the 64-column steps x 4 waves
        S, K = partial.shape[0], W.shape[1]
        steps = K // 256 if K % 256 == 0 else None
        for sp in range(S):
            if steps is not None and steps >= S:
                base, rem = divmod(steps, S)
                k0 = (sp * base + min(sp, rem)) * 256
                k1 = k0 + (base + (1 if sp < rem else 0)) * 256
            else:                 # (toy widths of the CPU tests: any partition of K will do)
                k0, k1 = sp * K // S, (sp + 1) * K // S
            partial[sp][rows] = x[:, k0:k1] @ W.float()[:, k0:k1].t()
        return partial
    acc = x @ W.float().t()
    if ln is not None:
        c1, c2, eps = ln
        mean = x.mean(1, keepdim=True)
        var = ((x * x).mean(1, keepdim=True) - mean * mean).clamp_min(0.0)
        rstd = torch.rsqrt(var + eps)
        acc = rstd * (acc - mean * c1.float()[None, :]) + c2.float()[None, :]
        if stats is not None:
            st = stats.view(-1, 2)
            st[rows, 0] = mean[:, 0]
            st[rows, 1] = rstd[:, 0]
    elif bias is not None:
        acc = acc + bias.float()
    if residual is not None:
        acc = acc + residual.float()[rows]
    if dact_in is not None:
        kind, pre = dact_in
        z = pre.detach().float()[rows].clone().requires_grad_(True)
        y = F.gelu(z) if kind == "gelu" else z * torch.sigmoid(1.702 * z)
        (d,) = torch.autograd.grad(y.sum(), z)
        acc = acc * d
    out[rows] = acc.to(out.dtype)
    if act_out is not None:
        kind, a = act_out
        a[rows] = (F.gelu(acc) if kind == "gelu" else acc * torch.sigmoid(1.702 * acc)).to(a.dtype)
    return out


def _part_width(N):
    return 80 if (N % 160 == 0 and N // 80 <= 16) else N // 2


def gemm_emits_parts(M, N, K, lora_rank_pad=0, W=None, dora=False):
    """(the emulation leaves partials for every even width - two per row where the 80-column tiling does not apply - so that the CPU tests
    exercise the producer / consumer plumbing on the tiny topologies too)"""
    return N // _part_width(N) if N % 2 == 0 else 0


def fold_layernorm(W, bias, gamma, beta, dtype=None):
    Wg = (W.float() * gamma.float()[None, :]).to(dtype or W.dtype).contiguous()
    c1 = Wg.float().sum(1).contiguous()
    c2 = (W.float() @ beta.float()).contiguous()
    if bias is not None:
        c2 = (c2 + bias.float()).contiguous()
    return Wg, c1, c2


def geglu_perm(H, device=None):
    j = torch.arange(H)
    pos_h = (j // 16) * 32 + j % 16
    perm = torch.empty(2 * H, dtype=torch.int64)
    perm[pos_h] = j
    perm[pos_h + 16] = H + j
    return perm


class GemmBatch:
    def __init__(self, items, device):
        self.items, self.n = items, len(items)


class DoraPlan:
    """torch emulation of ops.DoraPlan (sdlt_dora_refresh / _scale_wt / _mag_grad)."""

    def __init__(self, layers, wts, grads, rank, Rp, device):
        self.layers, self.wts, self.grads, self.rank = layers, wts, grads, rank

    def set_wts(self, wts):
        self.wts = wts

    def set_grads(self, grads):
        self.grads = grads

    def refresh(self, init=False):
        r = self.rank
        for L in self.layers:
            weff = L["W"].float() + L["s"] * L["B_s"][:, :r].float() @ L["A_s"][:r].float()
            nrm = weff.norm(dim=1)
            if init:
                L["mag"].copy_(nrm)
            L["scale"].copy_(L["mag"] / nrm)
            if L.get("Bt") is not None:
                L["Bt"].zero_()
                L["Bt"][:r].copy_((L["B32"] * L["scale"][:, None]).t().to(L["Bt"].dtype))
        self.scale_wts()

    def scale_wts(self):
        for w in self.wts:
            idx = torch.arange(w["src"].shape[1]) % w["period"]
            sc = torch.where(idx < w["nvalid"], w["scale"][idx.clamp(max=w["scale"].numel() - 1)], torch.zeros(()))
            w["dst"].copy_((w["src"].float() * sc).to(w["dst"].dtype))

    def mag_grad(self):
        for g in self.grads:
            N = g["Y"].shape[1]
            z = g["Y"].float() - (g["bias"].float() if g.get("bias") is not None else 0.0)
            gm = (g["dY"][:, :N].float() * z).sum(0) / g["mag"]
            if g.get("accumulate"):
                g["gmag"].add_(gm)
                continue
            g["gmag"].copy_(gm)
            g["gB"].mul_(g["scale"][:, None])


class LoraGradPlan:
    def __init__(self, problems, Rp, device):
        self.problems, self.accumulate = problems, False

    def set_accumulate(self, flag):
        self.accumulate = bool(flag)

    def run(self):
        for pr in self.problems:
            P, Q, out, R = pr["P"], pr["Q"], pr["out"], pr["R"]
            cv = pr.get("conv")
            if cv is None:
                Pm = P.float()
            else:  # im2col with k = tap*Cin + ci
                x = P.float().reshape(cv.B, cv.Hin, cv.Win, cv.Cin).permute(0, 3, 1, 2)
                cols = F.unfold(x, 3, padding=1, stride=cv.stride)            # [B, Cin*9, L] with (ci, tap) order
                cols = cols.reshape(cv.B, cv.Cin, 9, -1).permute(0, 3, 2, 1).reshape(-1, 9 * cv.Cin)
                Pm = cols
            g = Pm.t() @ Q.float()[:, :R]          # [Cw, R]
            g = g.t().contiguous() if pr["rank_major"] else g
            if self.accumulate:
                out.add_(g.reshape(out.shape))
            else:
                out.copy_(g.reshape(out.shape))


def _heads(t, B, Np, H, d):
    return t.float().reshape(B, Np, H, d).permute(0, 2, 1, 3)   # [B,H,Np,d]


def _attn_core(q, k, v, Nq, Nk, scale, causal):
    s = (q @ k.transpose(-1, -2)) * scale
    mask = torch.zeros(s.shape[-2], s.shape[-1], dtype=torch.bool)
    mask[:, Nk:] = True
    if causal:
        mask |= torch.triu(torch.ones_like(mask), diagonal=1)
    s = s.masked_fill(mask, float("-inf"))
    return s


def wsk_rowdot_shape(M, N, K, lora, d):
    return True          # (the emulation takes the side output everywhere, so that the CPU tests exercise the plumbing on the tiny topologies)


def attn_fwd(Q, K, V, Vt, O, L, *, B, H, Nq, Nk, Nqp, Nkp, d, scale, causal=False, zero_D=None):
    if zero_D is not None:
        zero_D.zero_()
    C = H * d
    q, k, v = _heads(Q[:, :C], B, Nqp, H, d), _heads(K[:, :C], B, Nkp, H, d), _heads(V[:, :C], B, Nkp, H, d)
    s = _attn_core(q, k, v, Nq, Nk, scale, causal)
    lse = torch.logsumexp(s, dim=-1)
    o = torch.softmax(s, dim=-1) @ v
    O[:, :C].copy_(o.permute(0, 2, 1, 3).reshape(B * Nqp, C).to(O.dtype))
    L.copy_(lse[:, :, :Nq].reshape(-1))
    return O


def attn_bwd(Q, K, V, Kt, Qt, O, L, dO, dOt, D, dQ, dK, dV, *, B, H, Nq, Nk, Nqp, Nkp, d, scale, causal=False,
             qsplit=1, dK32=None, dV32=None, accumulate_dq=False, accumulate_dk=False, defer_splitsum=False, d_ready=False):
    C = H * d
    if d_ready:      # the producer of dO left the row term: it must be what the pre-pass would have computed
        assert Nq == Nqp
        want = (dO[:, :C].float() * O[:, :C].float()).reshape(B, Nq, H, d).sum(-1).permute(0, 2, 1).reshape(-1)
        torch.testing.assert_close(D[: want.numel()], want, rtol=1e-4, atol=1e-5 * float(want.abs().max() + 1e-30))
    q = _heads(Q[:, :C], B, Nqp, H, d).detach().clone().requires_grad_(True)
    k = _heads(K[:, :C], B, Nkp, H, d).detach().clone().requires_grad_(True)
    v = _heads(V[:, :C], B, Nkp, H, d).detach().clone().requires_grad_(True)
    s = _attn_core(q, k, v, Nq, Nk, scale, causal)
    o = torch.softmax(s, dim=-1) @ v
    go = _heads(dO[:, :C], B, Nqp, H, d).clone()
    go[:, :, Nq:] = 0
    gq, gk, gv = torch.autograd.grad(o, [q, k, v], go)
    if defer_splitsum:        # dK / dV stay as fp32 slabs (all of it in slab 0 here); SplitsumPlan sums them into dK / dV later
        for slab, g in ((dK32, gk), (dV32, gv)):
            slab.zero_()
            slab[: B * Nkp, :C].copy_(g.permute(0, 2, 1, 3).reshape(B * Nkp, C))
        val = gq.permute(0, 2, 1, 3).reshape(B * Nqp, C)
        dQ[:, :C].copy_(((val + dQ[:, :C].float()) if accumulate_dq else val).to(dQ.dtype))
        return
    for dst, g, Np, acc in ((dQ, gq, Nqp, accumulate_dq), (dK, gk, Nkp, accumulate_dk), (dV, gv, Nkp, False)):
        val = g.permute(0, 2, 1, 3).reshape(B * Np, C)
        if acc:
            val = val + dst[:, :C].float()
        dst[:, :C].copy_(val.to(dst.dtype))


class SplitsumPlan:
    def __init__(self, items, device):
        self.items = items

    def run(self):
        for it in self.items:
            rows = it["B"] * it["Nkp"]
            Cw = it["dK"].shape[1]
            live = (torch.arange(rows) % it["Nkp"] < it["Nk"]).float()[:, None]
            for slab, out, acc in ((it["dK32"], it["dK"], it.get("acc0")), (it["dV32"], it["dV"], False)):
                val = slab[: it["nsplit"] * rows].reshape(it["nsplit"], rows, -1)[:, :, :Cw].sum(0) * live
                out.copy_(((val + out.float()) if acc else val).to(out.dtype))


def _gn_cat(x1, x2):
    return x1.float() if x2 is None else torch.cat([x1.float(), x2.float()], dim=1)


def groupnorm_fwd(x1, x2, y, stats, *, B, HW, gamma, beta, eps, silu, stats_zeroed=False):
    x = _gn_cat(x1, x2)
    C = x.shape[1]
    z = F.group_norm(x.reshape(B, HW, C).permute(0, 2, 1), 32, gamma.float(), beta.float(), eps)
    if silu:
        z = F.silu(z)
    y.copy_(z.permute(0, 2, 1).reshape(B * HW, C).to(y.dtype))
    return y


def groupnorm_colsum_splits(B, HW, Cc):
    return 2          # (the emulation always leaves two partial rows: the sums of the even and of the odd pixels)


class ColsumFinishPlan:
    def __init__(self, items, device):
        self.items = items

    def run(self):
        for ws, nsplit, out in self.items:
            out.copy_(ws[: nsplit * out.numel()].view(nsplit, -1).sum(0).to(out.dtype))


def groupnorm_bwd(x1, x2, dy, dx, stats, bstats, *, B, HW, gamma, beta, eps, silu, dres=None, stats_zeroed=False, colsum_ws=None):
    x = _gn_cat(x1, x2).detach().clone().requires_grad_(True)
    C = x.shape[1]
    z = F.group_norm(x.reshape(B, HW, C).permute(0, 2, 1), 32, gamma.float(), beta.float(), eps)
    if silu:
        z = F.silu(z)
    (g,) = torch.autograd.grad(z.permute(0, 2, 1).reshape(B * HW, C), x, dy.float())
    if dres is not None:
        g = g + dres.float()
    dx.copy_(g.to(dx.dtype))
    if colsum_ws is not None:      # [split][B][C] partial column sums of the stored rows
        r = dx.float().reshape(B, HW, C)
        colsum_ws[: 2 * B * C].view(2, B, C).copy_(torch.stack([r[:, 0::2].sum(1), r[:, 1::2].sum(1)]))
    return dx


def groupnorm_affine_grad(x1, x2, dy, stats, dgamma, dbeta, *, B, HW, gamma, beta, eps, silu, accumulate=False):
    x = _gn_cat(x1, x2).detach()
    C = x.shape[1]
    ga, be = gamma.detach().float().clone().requires_grad_(True), beta.detach().float().clone().requires_grad_(True)
    z = F.group_norm(x.reshape(B, HW, C).permute(0, 2, 1), 32, ga, be, eps)
    if silu:
        z = F.silu(z)
    g1, g2 = torch.autograd.grad(z.permute(0, 2, 1).reshape(B * HW, C), [ga, be], dy.float())
    if accumulate:
        dgamma.add_(g1)
        dbeta.add_(g2)
    else:
        dgamma.copy_(g1)
        dbeta.copy_(g2)


def layernorm_affine_grad(x, dy, stats, dgamma, dbeta, accumulate=False):
    xh = F.layer_norm(x.detach().float(), (x.shape[1],), None, None, 1e-5)
    g1, g2 = (dy.float() * xh).sum(0), dy.float().sum(0)
    if accumulate:
        dgamma.add_(g1)
        dbeta.add_(g2)
    else:
        dgamma.copy_(g1)
        dbeta.copy_(g2)


def wgrad_transpose(x, out, colsum_acc=None):
    out.zero_()
    out[:, :x.shape[0]] = x.t()
    if colsum_acc is not None:
        colsum_acc.add_(x.float().sum(0))
    return out


class AffineGradBatch:
    def __init__(self, items, device, *, groupnorm, B, HW, eps=0.0, silu=False):
        self.items, self.kw, self.gn = items, dict(B=B, HW=HW, eps=eps, silu=silu), groupnorm

    def run(self):
        for it in self.items:
            if self.gn:
                groupnorm_affine_grad(it["x1"], it["x2"], it["dy"], it["stats"], it["dgamma"], it["dbeta"], gamma=it["gamma"], beta=it["beta"],
                                      accumulate=True, **self.kw)
            else:
                layernorm_affine_grad(it["x1"], it["dy"], it["stats"], it["dgamma"], it["dbeta"], accumulate=True)


class WgradPanelBatch:
    def __init__(self, items, device, conv=None):
        self.items, self.n, self.conv = items, len(items), conv

    def run(self):
        for x, o, cs in self.items:
            if self.conv is None:
                wgrad_transpose(x, o, colsum_acc=cs)
            else:
                wgrad_im2col_t(x, o, **self.conv)


def wgrad_im2col_t(x, out, *, B, H, W, stride=1, ups=1):
    C = x.shape[1]
    img = x.float().reshape(B, H, W, C).permute(0, 3, 1, 2)
    if ups == 2:
        img = F.interpolate(img, scale_factor=2, mode="nearest")
    cols = F.unfold(img, 3, padding=1, stride=stride)                  # [B, C*9, L], row c*9 + tap
    L = cols.shape[-1]
    cols = cols.reshape(B, C, 9, L).permute(2, 1, 0, 3).reshape(9 * C, B * L)     # row tap*C + c, column (b, oy, ox)
    out.zero_()
    out[:, :B * L] = cols.to(out.dtype)
    return out


def layernorm_fwd(x, y, stats, *, gamma, beta, eps=1e-5):
    y.copy_(F.layer_norm(x.float(), (x.shape[1],), gamma.float(), beta.float(), eps).to(y.dtype))
    return y


def layernorm_bwd(x, dy, dx, stats, *, gamma, dres=None, dy_slabs=None, beta=None, y_out=None):
    if dy_slabs is not None:
        dy = dy_slabs.float().sum(0)
    if y_out is not None:     # sdlt_layernorm_bwd_y: the normalised rows as a second output
        y_out.copy_(F.layer_norm(x.float(), (x.shape[1],), gamma.float(), beta.float(), 1e-5).to(y_out.dtype))
    xx = x.detach().float().clone().requires_grad_(True)
    z = F.layer_norm(xx, (x.shape[1],), gamma.float(), None, 1e-5)
    (g,) = torch.autograd.grad(z, xx, dy.float())
    if dres is not None:
        g = g + dres.float()
    dx.copy_(g.to(dx.dtype))
    return dx


def geglu_fwd(inp, out):
    h, g = inp.float().chunk(2, dim=1)
    out.copy_((h * F.gelu(g)).to(out.dtype))
    return out


def geglu_bwd(inp, dout, din):
    x = inp.detach().float().clone().requires_grad_(True)
    h, g = x.chunk(2, dim=1)
    (gr,) = torch.autograd.grad(h * F.gelu(g), x, dout.float())
    din.copy_(gr.to(din.dtype))
    return din


def map_bf16(op, x, dy, y):
    xx = x.float()
    if op == MAP_SILU:
        r = F.silu(xx)
    elif op == MAP_ADD:
        r = xx + dy.float()
    elif op == MAP_GELU:
        r = F.gelu(xx)
    elif op == MAP_QGELU:
        r = xx * torch.sigmoid(1.702 * xx)
    else:
        xx = xx.detach().clone().requires_grad_(True)
        f = {MAP_DSILU: F.silu, MAP_DGELU: F.gelu, MAP_DQGELU: lambda t: t * torch.sigmoid(1.702 * t)}[op](xx)
        (r,) = torch.autograd.grad(f, xx, dy.float())
    y.copy_(r.to(y.dtype))
    return y


def timestep_embedding(t, out):
    half = out.shape[1] // 2
    f = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=F32) / half)
    a = t.float()[:, None] * f[None]
    out.copy_(torch.cat([torch.cos(a), torch.sin(a)], dim=1).to(out.dtype))
    return out


def add_noise_nhwc(x0, noise, timesteps, alphas_cumprod, out, noisy_nchw=None):
    B, C, H, W = x0.shape
    a = alphas_cumprod[timesteps].view(B, 1, 1, 1)
    noisy = a.sqrt() * x0 + (1 - a).sqrt() * noise
    if noisy_nchw is not None:
        noisy_nchw.copy_(noisy)
    out.zero_()
    out[:, :C].copy_(noisy.permute(0, 2, 3, 1).reshape(B * H * W, C).to(out.dtype))
    return out


def masked_mse_fwd_bwd(pred, noise, noisy, mask, timesteps, alphas_cumprod, sums, loss_out, dpred, *, snr_gamma,
                       v_prediction=False, loss_scale=1.0):
    B, C, H, W = noise.shape
    p = pred[:, :C].detach().float().reshape(B, H, W, C).permute(0, 3, 1, 2).clone().requires_grad_(True)
    a = alphas_cumprod[timesteps].view(B, 1, 1, 1)
    target = a.sqrt() * noise - (1 - a).sqrt() * noisy if v_prediction else noise
    e = (p - target).pow(2) * mask
    per = e.flatten(1).mean(1)
    if snr_gamma:
        snr = (alphas_cumprod[timesteps].sqrt() / (1 - alphas_cumprod[timesteps]).sqrt()) ** 2
        w = torch.minimum(snr, torch.full_like(snr, float(snr_gamma))) / snr + (1.0 if v_prediction else 0.0)
        loss = (per * (w / w.mean())).mean()
    else:
        mm = mask.flatten(1).mean(1)
        loss = (per / (mm / mm.mean())).mean()
    (g,) = torch.autograd.grad(loss * loss_scale, p)
    loss_out.fill_(float(loss.detach()))
    dpred.zero_()
    dpred[:, :C].copy_(g.permute(0, 2, 3, 1).reshape(B * H * W, C).to(dpred.dtype))


def adamw_fused(p, g, m, v, hyper, l1_sum=None):
    lr, b1, b2, eps, wd, bc1, bc2, l1c, gs = [float(x) for x in hyper[:9]]
    if l1_sum is not None:
        l1_sum.fill_(float(p.abs().sum()))
    gi = g * gs + l1c * torch.sign(p)
    m.mul_(b1).add_(gi, alpha=1 - b1)
    v.mul_(b2).addcmul_(gi, gi, value=1 - b2)
    p.mul_(1 - lr * wd).addcdiv_(m, v.sqrt() / math.sqrt(bc2) + eps, value=-lr / bc1)


PRODIGY_HYPER = ("lr", "beta1", "beta2", "beta3", "eps", "weight_decay", "d_coef", "growth_rate", "l1_coef", "grad_scale",
                 "use_bias_correction", "safeguard_warmup", "decouple")
PRODIGY_STATE = ("d", "d0", "d_max", "d_numerator", "d_denom", "d_hat", "k", "dlr", "active")


def prodigy_step(p, g, p0, m, v, s, hyper, state, acc, l1_sum=None):
    """sdlt_prodigy_step in torch (fp32 tensors, fp64 sums): the same four stages as csrc/optim.hip."""
    h = dict(zip(PRODIGY_HYPER, [float(x) for x in hyper[:len(PRODIGY_HYPER)]]))
    st = dict(zip(PRODIGY_STATE, [float(x) for x in state[:len(PRODIGY_STATE)]]))
    f32 = lambda x: float(torch.tensor(x, dtype=torch.float32))  # noqa: E731
    bc = 1.0
    if h["use_bias_correction"]:
        bc = f32(math.sqrt(f32(1 - f32(h["beta2"] ** (st["k"] + 1)))) / f32(1 - f32(h["beta1"] ** (st["k"] + 1))))
    d, d0, dlr = st["d"], st["d0"], f32(f32(st["d"] * h["lr"]) * bc)
    if l1_sum is not None:
        l1_sum.fill_(float(p.abs().sum()))
    dot = den = 0.0
    if h["lr"] > 0:
        gi = g * h["grad_scale"] + h["l1_coef"] * torch.sign(p)
        if not h["decouple"] and h["weight_decay"] != 0:
            gi = gi + h["weight_decay"] * p
        dot = float((gi.double() * (p0 - p).double()).sum())
        m.mul_(h["beta1"]).add_(gi, alpha=f32(d * f32(1 - h["beta1"])))
        v.mul_(h["beta2"]).addcmul_(gi, gi, value=f32(f32(d * d) * f32(1 - h["beta2"])))
        s.mul_(h["beta3"]).add_(gi, alpha=f32(f32(d / d0) * (d if h["safeguard_warmup"] else dlr)))
        den = float(s.abs().double().sum())
    state[PRODIGY_STATE.index("dlr")] = dlr
    if den == 0.0:
        state[PRODIGY_STATE.index("active")] = 0.0
        return
    num = st["d_numerator"] * h["beta3"] + f32(d / d0) * dlr * dot
    d_hat, d_max = d, st["d_max"]
    if h["lr"] > 0:
        d_hat = f32(h["d_coef"] * num / den)
        if d == d0:
            d = max(d, d_hat)
        d_max = max(d_max, d_hat)
        d = min(d_max, f32(d * h["growth_rate"]))
    for k_, val in (("d_numerator", num), ("d_denom", den), ("d", d), ("d_max", d_max), ("d_hat", d_hat), ("k", st["k"] + 1), ("active", 1.0)):
        state[PRODIGY_STATE.index(k_)] = val
    decay = 1.0 - h["weight_decay"] * dlr if h["decouple"] else 1.0
    p.mul_(decay).addcdiv_(m, v.sqrt() + f32(d * h["eps"]), value=-dlr)


class LnFoldPlan:
    """sdlt_ln_fold_adapters: Ag = A o gamma (rounded), cA = rowsum(Ag), abeta = A beta."""

    def __init__(self, items, device):
        self.items = items

    def run(self):
        for it in self.items:
            A, Ag, c = it["A32"].float(), it["Ag"], it["consts"]
            r = A.shape[0]
            Ag.zero_()
            Ag[:r].copy_((A * it["gamma"].float()[None, :]).to(Ag.dtype))
            c.zero_()
            c[:r].copy_(Ag[:r].float().sum(1))
            c[16:16 + r].copy_(A @ it["beta"].float())


class ShadowPlan:
    def __init__(self, entries, device):
        self.entries = entries

    def run(self, arena):
        for (off, rows, cols, src_ld, dst, dstT) in self.entries:
            src = torch.as_strided(arena, (rows, cols), (src_ld, 1), off)
            if dst is not None:
                dst[:rows, :cols].copy_(src.to(dst.dtype))
            if dstT is not None:
                dstT[:cols, :rows].copy_(src.t().to(dstT.dtype))


def _shadow_adamw(self, p, g, m, v, hyper):
    """sdlt_adamw_shadow_refresh: AdamW on exactly the elements the descriptors tile, then the refresh."""
    lr, b1, b2, eps, wd, bc1, bc2, _, gs = [float(x) for x in hyper[:9]]
    for (off, rows, cols, src_ld, dst, dstT) in self.entries:
        views = [torch.as_strided(t, (rows, cols), (src_ld, 1), off) for t in (p, g, m, v)]
        pv, gv, mv, vv = views
        gi = gv * gs
        mv.mul_(b1).add_(gi, alpha=1 - b1)
        vv.mul_(b2).addcmul_(gi, gi, value=1 - b2)
        pv.mul_(1 - lr * wd).addcdiv_(mv, vv.sqrt() / math.sqrt(bc2) + eps, value=-lr / bc1)
    self.run(p)


ShadowPlan.adamw = _shadow_adamw


def q8_tables(device):
    from oracle import adam8bit_ref as A8
    parts = []
    for q in A8.maps():
        parts += [q, torch.cat([(q[:-1] + q[1:]) * 0.5, torch.tensor([float("inf")])])]
    return torch.cat(parts).to(device)


def _shadow_adamw8(self, p, g, m8, v8, absmax, tables, hyper):
    """sdlt_adamw8_shadow_refresh through the oracle's AdamW8bit (same block shape and absmax layout), then the refresh."""
    from oracle import adam8bit_ref as A8
    lr, b1, b2, eps, wd, bc1, bc2, _, gs = [float(x) for x in hyper[:9]]
    step = max(1, round(math.log(max(1.0 - bc1, 1e-300)) / math.log(b1)))
    nb = 0
    for (off, rows, cols, src_ld, dst, dstT) in self.entries:
        pv, gv = [torch.as_strided(t, (rows, cols), (src_ld, 1), off) for t in (p, g)]
        tr, tc = (rows + 63) // 64, (cols + 63) // 64
        blocks = tr * tc
        # codes are tile-major: [blocks][64][64] -> the [rows, cols] matrix and back
        from_tiles = lambda t: t[4096 * nb: 4096 * (nb + blocks)].view(tr, tc, 64, 64).permute(0, 2, 1, 3).reshape(tr * 64, tc * 64)[:rows, :cols].clone()  # noqa: E731
        st = A8.Adam8State(rows, cols)
        st.m8, st.v8 = from_tiles(m8), from_tiles(v8)
        st.set_tile_absmax(absmax[4 * nb: 4 * (nb + blocks)])
        pv.copy_(A8.adamw8_step(pv.clone(), gv, st, lr=lr, beta1=b1, beta2=b2, eps=eps, weight_decay=wd, step=step, grad_scale=gs))
        for dst8, src8 in ((m8, st.m8), (v8, st.v8)):
            full = torch.zeros(tr * 64, tc * 64, dtype=torch.uint8)
            full[:rows, :cols] = src8
            dst8[4096 * nb: 4096 * (nb + blocks)] = full.view(tr, 64, tc, 64).permute(0, 2, 1, 3).reshape(-1)
        absmax[4 * nb: 4 * (nb + blocks)] = st.tile_absmax().reshape(-1)
        nb += blocks
    self.run(p)


ShadowPlan.adamw8 = _shadow_adamw8


def adamw8_flat(p, g, m8, v8, absmax, tables, hyper):
    """sdlt_adamw8_flat through the oracle: the range as a [n / 64, 64] matrix - a block of 32 rows x 64 columns IS 2048 consecutive elements."""
    from oracle import adam8bit_ref as A8
    lr, b1, b2, eps, wd, bc1, bc2, _, gs = [float(x) for x in hyper[:9]]
    step = max(1, round(math.log(max(1.0 - bc1, 1e-300)) / math.log(b1)))
    n = p.numel()
    rows = (n + 63) // 64
    pad = lambda t, dt, fill=0: torch.cat([t.reshape(-1).to(dt), torch.full((rows * 64 - n,), fill, dtype=dt)]).view(rows, 64)  # noqa: E731
    st = A8.Adam8State(rows, 64)
    st.m8, st.v8 = pad(m8, torch.uint8, 127), pad(v8, torch.uint8)          # (the padding decodes to 0: code 127 of the signed book, code 0 of the unsigned one)
    nb = (n + 2047) // 2048
    st.am, st.av = absmax.view(nb, 2)[:, 0:1].clone(), absmax.view(nb, 2)[:, 1:2].clone()
    gp = pad(g, torch.float32)
    pn = A8.adamw8_step(pad(p, torch.float32), gp, st, lr=lr, beta1=b1, beta2=b2, eps=eps, weight_decay=wd, step=step, grad_scale=gs)
    p.copy_(pn.reshape(-1)[:n])
    m8.copy_(st.m8.reshape(-1)[:n]), v8.copy_(st.v8.reshape(-1)[:n])
    absmax.view(nb, 2)[:, 0:1].copy_(st.am)
    absmax.view(nb, 2)[:, 1:2].copy_(st.av)
ShadowPlan.n_blocks = property(lambda self: sum(((r + 63) // 64) * ((c + 63) // 64) for (_, r, c, _, _, _) in self.entries))


def add2d(a, b, out):
    out.copy_((a.float() + b.float()).to(out.dtype))
    return out


def sum2x2(inp, out, *, B, H, W):
    C = inp.shape[1]
    x = inp.float().reshape(B, H, 2, W, 2, C).sum(dim=(2, 4))
    out.copy_(x.reshape(B * H * W, C).to(out.dtype))
    return out


def colsum(x, out, *, B, R):
    out.copy_(x.float().reshape(B, R, -1).sum(1).to(out.dtype))
    return out


def embed_gather(table, ids, pos, out, *, B, T, Tp):
    out.zero_()
    emb = table.float()[ids.reshape(B, T)] + pos.float()[:T][None]
    out.view(B, Tp, -1)[:, :T].copy_(emb.to(out.dtype))
    return out


def embed_grad(dx, ids, train_ids, grad, *, B, T, Tp, accumulate=False):
    d = dx.float().reshape(B, Tp, -1)[:, :T]
    g = torch.stack([(d * (ids.reshape(B, T) == int(t))[..., None]).sum(dim=(0, 1)) for t in train_ids])
    grad.copy_(grad + g if accumulate else g)
    return grad


def ti_std_reg(rows, grad, loss_out, *, target_mean, target_var, weight):
    r = rows.detach().clone().requires_grad_(True)
    loss = weight * ((target_mean - r.std(-1)) ** 2 / target_var).mean()
    (g,) = torch.autograd.grad(loss, r)
    grad.add_(g)
    loss_out.add_(float(loss))
