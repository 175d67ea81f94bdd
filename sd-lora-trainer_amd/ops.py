"""Tensor-level wrappers over the C-ABI (include/sdlt_kernels.h).  PyTorch is plumbing here: it owns
the device memory and the stream; all arithmetic happens in libsdlt_kernels.so.

Conventions: activations are 2-D bf16 matrices [rows, C] with unit column stride and an
arbitrary row stride (views of wider buffers are fine); "NHWC" means rows = B*H*W.
"""
import ctypes as C
import os
import weakref

import torch

from . import _lib

BF16 = torch.bfloat16
F32 = torch.float32


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _ld(t):
    assert t.dim() == 2 and t.stride(1) == 1, f"need a 2-D row-major view, got {tuple(t.shape)} strides {t.stride()}"
    return t.stride(0)


def _chk2(t, dtype=BF16):
    assert t.is_cuda and t.dtype == dtype, f"expected cuda {dtype}, got {t.device} {t.dtype}"
    return t


_zero_pages = {}


def zero_page(device):
    z = _zero_pages.get(device)
    if z is None:
        z = torch.zeros(256, dtype=BF16, device=device)
        _zero_pages[device] = z
    return z


_workspaces = {}
_ws_owner = None
_gn_owner = None


class norm_workspace_owner:
    """`with norm_workspace_owner(key):` - the GroupNorm / column-sum reductions enqueued inside use `key`'s scratch.  These only run on a
    job's MAIN stream, so the scope is set for every job unconditionally - also for jobs whose text encoders fork onto side streams and
    therefore keep per-stream split-K workspaces (two such jobs captured on torch's shared capture stream would otherwise bake the same
    partial-sum rows into both graphs)."""

    def __init__(self, key):
        self.key = key

    def __enter__(self):
        global _gn_owner
        self.prev, _gn_owner = _gn_owner, self.key

    def __exit__(self, *a):
        global _gn_owner
        _gn_owner = self.prev


class workspace_owner:
    """`with workspace_owner(key):` - every GEMM enqueued inside uses the split-K workspace of `key` instead of the current
    stream's.  A training job captures its step under its own key: two jobs whose graphs replay concurrently on one GPU must
    not share slabs / counters, and torch may capture both graphs on the same internal stream."""

    def __init__(self, key):
        self.key = key

    def __enter__(self):
        global _ws_owner
        self.prev, _ws_owner = _ws_owner, self.key

    def __exit__(self, *a):
        global _ws_owner
        _ws_owner = self.prev


def splitk_workspace(device):
    """Split-K scratch of sdlt_gemm_bf16: fp32 partial-tile slabs + zero-initialised arrival counters.
    One workspace per (device, stream) - or per owner, see workspace_owner: GEMMs enqueued on different streams may run
    concurrently (the two text encoders do) and must not share slabs or counters; GEMMs on one stream are ordered, so they can."""
    key = (device, ("owner", _ws_owner) if _ws_owner is not None else torch.cuda.current_stream(device).cuda_stream)
    ws = _workspaces.get(key)
    if ws is None:
        ws = (torch.empty(96 << 20, dtype=torch.uint8, device=device), torch.zeros(4096, dtype=torch.int32, device=device))
        _workspaces[key] = ws
    return ws


class ConvGeom:
    """Geometry of an implicit 3x3 convolution over an NHWC activation (sdlt_gemm_bf16 mode 1)."""
    __slots__ = ("B", "Hin", "Win", "Cin", "Hout", "Wout", "stride", "ups", "flip", "tr")

    def __init__(self, B, Hin, Win, Cin, Hout, Wout, stride=1, ups=1, flip=0, tr=0):
        self.B, self.Hin, self.Win, self.Cin, self.Hout, self.Wout = B, Hin, Win, Cin, Hout, Wout
        self.stride, self.ups, self.flip, self.tr = stride, ups, flip, tr


THROUGHPUT_HINT = False


def set_throughput_hint(flag):
    """Several independent jobs are stepped concurrently on this device (train.train_concurrent, bench.py --jobs-per-gpu): every
    GEMM enqueued (i.e. captured) from now on carries sdlt_gemm_params.throughput_hint."""
    global THROUGHPUT_HINT
    # Round 4: with the wave-split-K kernel and the folded LayerNorm in the one-job plan the hint LOSES (two SDXL jobs: 72.3 ms per pair of steps with it,
    # 68.5 without, same session) - concurrent jobs keep the one-job kernel choices unless SDLT_THROUGHPUT_HINT=1 asks for the round-2 rules.
    THROUGHPUT_HINT = bool(flag) and os.environ.get("SDLT_THROUGHPUT_HINT", "0") == "1"


WSK = os.environ.get("SDLT_WSK", "1") != "0"


WSK_LORA = os.environ.get("SDLT_WSK_LORA", "1") != "0"


WSK_KMAX = int(os.environ.get("SDLT_WSK_KMAX", "10240"))      # longest plain K on the wave-split-K kernel (whole-step A/B: 10240 -> -0.14 ms against 8192)


def wsk_shape(M, N, K, lora=False):
    """Shapes the wave-split-K kernel takes over from the tiled one: 64 x 80 tiles that fill the 256 CUs exactly once (or less).  Without an
    adapter: a K long enough for its flatter per-step cost to pay (tools/wsk_probe.py: 1024 x 1280, K = 3840 / 5120: 19.8 / 23.3 us against 28.4 /
    31.1 us; equal at K = 1280 and, in the probe, at 10240 - in the step the dX of ff.net.0 (K = 10240) is 2 us faster here; slower on wider or taller outputs, which make more than 256 tiles).  With a rank-16 adapter: the
    1024 x 1280 x 1280 projections (to_q / to_out.0 and their input gradients, 310 launches of an SDXL step)."""
    if not (M % 64 == 0 and N % 640 == 0 and K % 256 == 0 and (M // 64) * (N // 80) <= 256):
        return False
    if lora:
        return WSK_LORA and 1024 <= K <= 8192 and (M // 64) * (N // 80) >= 128
    return 2560 <= K <= WSK_KMAX


ROWDOT = os.environ.get("SDLT_ROWDOT", "1") != "0"          # the attention backward's row term D from the GEMM that produces dO (A/B switch)


def wsk_rowdot_shape(M, N, K, lora, d):
    """Will gemm(rowdot=) apply to the dX of a to_out.0 of this shape?  (The caller only uses it to decide whether the forward zeroes D; gemm() itself reports what it did.)"""
    return ROWDOT and WSK and not THROUGHPUT_HINT and d == 64 and N % 64 == 0 and wsk_shape(M, N, K, lora)


WSK_PACK = os.environ.get("SDLT_WSK_PACK", "1") != "0"
WSK_DORA = os.environ.get("SDLT_WSK_DORA", "1") != "0"        # DoRA's column factor in the wave-split-K epilogue (A/B switch: 0 = the tiled kernel, round 5)
WSK_RANKS = tuple(int(x) for x in os.environ.get("SDLT_WSK_RANKS", "16,32").split(","))      # padded adapter ranks the wave-split-K kernel takes (A/B switch: "16" = round 5; 64 does not exist: csrc/wsk.hip says why)
_WSK_FROZEN = {}         # data_ptr of a weight declared frozen -> weakref of the tensor
_WSK_PACKED = {}         # data_ptr -> (fragment-major copy (sdlt_wsk_pack_weight's layout), N, K, ld, weakref of the tensor, W._version at pack time)


def _wsk_forget(key, ref):
    """weakref callback of a frozen weight: the tensor died - drop its registration and its packed copy (1-2 GB per SDXL model) with it, unless the
    address already belongs to a newer registration."""
    f = _WSK_FROZEN.get(key)
    if f is ref or (f is not None and f() is None):
        _WSK_FROZEN.pop(key, None)
    e = _WSK_PACKED.get(key)
    if e is not None and (e[4] is ref or e[4]() is None):
        _WSK_PACKED.pop(key, None)


def wsk_mark_frozen(W):
    """Declare a bf16 weight [N, K] FROZEN: nothing rewrites it for as long as the tensor lives (the UNet under LoRA / textual inversion; NOT the full
    fine-tune's refreshed operands, NOT DoRA's per-step W^T).  The first gemm(X, W, ...) that runs on the wave-split-K kernel then makes the
    fragment-major copy of include/sdlt_kernels.h (sdlt_wsk_pack_weight) once and every such call reads the weight through it.  The copy lives as
    long as the tensor (a weakref callback drops it); code that DOES rewrite a marked weight in place (loading a checkpoint into the same tensors,
    merging adapters) calls wsk_invalidate(W) - eager calls also notice the tensor's version counter and re-pack, a captured graph cannot."""
    if WSK and WSK_PACK and W is not None and W.dim() == 2 and W.dtype == BF16 and W.is_cuda and W.stride(1) == 1:
        key = W.data_ptr()
        e = _WSK_PACKED.get(key)
        if e is not None and e[4]() is not W:          # a copy made for an earlier tensor at this address
            del _WSK_PACKED[key]
        _WSK_FROZEN[key] = weakref.ref(W, lambda r, key=key: _wsk_forget(key, r))


def wsk_invalidate(W=None):
    """Drop the packed copy of one marked weight (it is re-made on the next eager use) or, with no argument, of all of them; call it after writing a
    marked weight in place and before re-capturing a graph that reads it (a captured launch holds the OLD copy's pointer)."""
    if W is None:
        _WSK_PACKED.clear()
    else:
        _WSK_PACKED.pop(W.data_ptr(), None)


def _wsk_is_frozen(W):
    """Is W a weight wsk_mark_frozen() knows (so that _wsk_operand hands out its packed copy)?"""
    f = _WSK_FROZEN.get(W.data_ptr())
    return f is not None and f() is not None and tuple(f().shape) == tuple(W.shape) and f().stride(0) == W.stride(0)


def _wsk_operand(W):
    """(pointer, ld) of W for sdlt_wsk_gemm*: the packed copy with ld 0 for a weight declared frozen."""
    key = W.data_ptr()
    e = _WSK_PACKED.get(key)
    if e is not None:
        if e[4]() is None or e[5] != W._version:      # the registered tensor is gone (its address may have been reused), or it was rewritten in place: drop the copy
            del _WSK_PACKED[key]
        elif e[1] == W.shape[0] and e[2] == W.shape[1] and e[3] == _ld(W):
            return _p(e[0]), 0
    f = _WSK_FROZEN.get(key)
    if f is not None and f() is None:
        del _WSK_FROZEN[key]
        f = None
    if f is not None and tuple(f().shape) == tuple(W.shape) and f().stride(0) == W.stride(0) and key not in _WSK_PACKED:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("wave-split-K GEMM: the packed copy of a frozen weight is made on its first eager use - run the body once before capturing it")
        N, K = W.shape
        Wp = torch.empty(N * K, device=W.device, dtype=BF16)
        _lib.check(_lib.load().sdlt_wsk_pack_weight(_p(W), _ld(W), N, K, _p(Wp), _stream()), "sdlt_wsk_pack_weight")
        _WSK_PACKED[key] = (Wp, N, K, _ld(W), f, W._version)
        return _p(Wp), 0
    return _p(W), _ld(W)


# ---- next-weight prefetch (round 6): a wave-split-K launch is bound by the fabric's first-touch burst of its operands; the launch in front of it can touch the
# first K steps of its packed weight into the L2s that will read them (sdlt_wsk_gemm_params.pf_next_w).  The plan is static, so the "next weight" of every launch is known from
# one eager pass: step.TrainStep records the sequence of wave-split-K products of each graph it captures (pf_record_*) and replays it while capturing (pf_replay_*); a launch
# whose (weight, N, K) differs from the recorded pass switches the hints off for the rest of that capture.  Hints never change a result.  SDLT_WSK_PREFETCH=0: off (A/B).
WSK_PREFETCH = os.environ.get("SDLT_WSK_PREFETCH", "1") != "0"
WSK_PREFETCH_STEPS = int(os.environ.get("SDLT_WSK_PREFETCH_STEPS", "40"))      # K steps of 64 columns touched per column tile (tools/wsk_prefetch_probe.py)
_PF = {"mode": None, "seq": None, "idx": 0}


def pf_record_begin():
    _PF.update(mode="record", seq=[], idx=0)


def pf_record_end():
    seq = _PF["seq"]
    _PF.update(mode=None, seq=None, idx=0)
    return seq


def pf_replay_begin(seq):
    _PF.update(mode="replay" if (WSK_PREFETCH and seq) else None, seq=seq, idx=0)


def pf_replay_end():
    _PF.update(mode=None, seq=None, idx=0)


def _pf_hint(Wptr, Wld, N, K):
    """The (pointer, N, K) of the packed weight the NEXT wave-split-K product of the recorded plan reads, or None."""
    m = _PF["mode"]
    if m is None:
        return None
    ent = ((Wptr.value or 0) if Wld == 0 else 0, int(N), int(K))
    if m == "record":
        _PF["seq"].append(ent)
        return None
    seq, i = _PF["seq"], _PF["idx"]
    if i >= len(seq) or seq[i] != ent:          # not the plan that was recorded: no hints from here on
        _PF["mode"] = None
        return None
    _PF["idx"] = i + 1
    if i + 1 < len(seq) and seq[i + 1][0] and seq[i + 1][0] != ent[0] and seq[i + 1][1] % 320 == 0:
        return seq[i + 1]
    return None


WSK_CONV = os.environ.get("SDLT_WSK_CONV", "1") != "0"


def wsk_conv_shape(conv, N, lora_rank_pad=0):
    """3 x 3 convolutions the wave-split-K kernel takes over from the tiled one (sdlt_wsk_conv): stride 1 at the resolution where 64 x 80 output tiles fill the
    256 CUs at most once - the 32 x 32 level of SDXL at 1024 px (ResnetBlock2D conv1 / conv2 of the 1280-wide stages, forward and input gradient)."""
    if not (WSK and WSK_CONV) or conv.stride != 1 or conv.ups != 1 or conv.tr or conv.Hin != conv.Hout or conv.Win != conv.Wout:
        return False
    M = conv.B * conv.Hout * conv.Wout
    if M % 64 or N % 640 or conv.Cin % 64 or lora_rank_pad not in (0, 16):
        return False
    # (probed, tools/wsk_conv_probe.py: 256 tiles - 1280 -> 1280 62 -> 52 us, 640 -> 1280 38.5 -> 28, 2560 -> 1280 93 -> 91; the 128 tiles of a 640-wide output LOSE, 42 -> 47 us)
    return 200 <= (M // 64) * (N // 80) <= 256


def gemm_emits_parts(M, N, K, lora_rank_pad=0, W=None, dora=False):
    """Number of row partials per row (0: none) a plain / rank-16-adapter product [M, K] x [N, K]^T (+ bias, residual) runs on the wave-split-K kernel and can therefore leave row
    partials for the next LayerNorm (gemm(..., ln_parts_out=)): the to_out.0 and ff.net.2 products of the 1280-wide blocks at batch 1."""
    ok = (WSK and not THROUGHPUT_HINT and (WSK_DORA or not dora) and (lora_rank_pad in (0, 16) or (lora_rank_pad in WSK_RANKS and WSK_PACK and W is not None and _wsk_is_frozen(W))) and wsk_shape(M, N, K, lora_rank_pad > 0)
          and (N // 80) % 2 == 0 and N // 80 <= 16)
    return N // 80 if ok else 0


def gemm(X, W, out, *, X2=None, W2=None, conv=None, lora=None, bias=None, rowbias=None, rows_per_batch=0,
         residual=None, alpha=1.0, Ct=None, tile=0, splitk=0, stages=0, accumulate=False, lora_group_n=0, lora_group_k=0, batch=None,
         geglu_out=None, geglu_bwd=None, act_out=None, dact_in=None, col_scale=None, ln=None, ln_parts_out=None, rowdot=None, out0=None):
    """out[M,N] = alpha*col_scale[n]*(X.W^T [+ X2.W2^T] [+ s*(X.Adown^T).Bup^T]) + bias + rowbias[m//rows_per_batch] + residual.
    out0 [M,N] (with residual, adapter launches): ALSO the value before the residual (DoRA: the magnitude gradient reads the layer's own output) - one launch where the
    product runs on the wave-split-K kernel (sdlt_wsk_gemm_params.Y0), otherwise the product into out0 and an add2d launch.
    col_scale fp32 [N]: DoRA's magnitude / norm factor (adapter launches only; DoraPlan keeps it up to date).
    act_out = (kind, A [M,N]): also writes A = act(out), kind "gelu" | "quick_gelu" (the CLIP MLP's fc1).
    dact_in = (kind, P [M,N]): out = (...) * act'(P), P = the forward pre-activation (the dX of the CLIP MLP's fc2).
    geglu_out [M, N/2]: this GEMM is ff.net.0.proj in the interleaved-16 layout (geglu_perm) - also writes hidden * gelu(gate).
    geglu_bwd = (F1 [M, 2N], dF1 [M, 2N]): this GEMM is the dX of ff.net.2 - writes GEGLU's input gradient instead of `out` (None).
    lora = (Adown [Rp,K], Bup [N,Rp], scale, T_out [M,Rp] or None).  out dtype bf16 or fp32.  Ct: optional
    transposed bf16 copy [N, >=M].  conv: ConvGeom -> X is the NHWC activation [B*Hin*Win, Cin].
    lora_group_n > 0: W is a stack of G = N / lora_group_n projections with one adapter each:
    Adown [G*Rp, K] (group-major), Bup [N, Rp], T_out [M, G*Rp].
    lora_group_k > 0: X is a stack of G = K / lora_group_k gradients (the dX of such a stack): Adown [Rp, K], Bup [N, G*Rp],
    T_out [M, G*Rp].
    ln = (c1 fp32 [N], stats fp32 [M, 2] or None, eps, adapter constants fp32 [G*32] or None): the LayerNorm in front of this product is
    folded in (fold_layernorm: X = the raw rows, W = W o gamma, bias = c2; with an adapter Adown = A o gamma, LnFoldPlan) - sdlt_gemm_params.ln_c1.
    ... optionally followed by (parts fp32 [M, P, 2], P): the row partials the PRODUCER of X left (ln_parts_out of that call).
    ln_parts_out fp32 [M, N / 80, 2]: also leave the row partials (sum, centred sum of squares of the rounded output row per 80-column tile) for the
    LayerNorm that reads `out` next - only where this call runs on the wave-split-K kernel (gemm_emits_parts says so; asserted).
    rowdot = dict(O bf16 [M, N], D fp32 [B * N / 64 * Nq], Nq): `out` is the gradient dO of a self-attention with 64-wide heads; where this call runs on the wave-split-K
    kernel it also accumulates D[b, h, q] += sum_n rounded(out[m, n]) O[m, n] over each head's columns and sets rowdot["done"] = True (attn_bwd(d_ready=True) then skips its
    D pre-pass); otherwise the dict is left alone and the caller keeps the pre-pass.  D must be zero on entry (attn_fwd(zero_D=)).
    batch: a GemmBatch - the launch runs len(batch) problems of identical shape / leading dimensions; the tensor arguments
    describe problem 0 (shapes, strides, options), every problem's operand pointers come from the batch."""
    lib = _lib.load()
    rp_ = lora[0].shape[0] if lora is not None else 0
    if out0 is not None:
        assert residual is not None and lora is not None and Ct is None and ln is None and geglu_out is None and act_out is None and tuple(out0.shape) == tuple(out.shape)
    if (WSK and conv is None and X2 is None and rowbias is None and alpha == 1.0 and Ct is None and batch is None and geglu_out is None
            and geglu_bwd is None and act_out is None and dact_in is None and (col_scale is None or (lora is not None and ln is None and WSK_DORA)) and not accumulate and tile == 0 and splitk == 0
            and not lora_group_n and out is not None and out.dtype == BF16 and not THROUGHPUT_HINT and (ln is None or not lora_group_k)
            and (lora is None or (rp_ == 16 and ((not lora_group_k and lora[1].shape[1] == 16) or
                                                 (lora_group_k and lora_group_k % 64 == 0 and W.shape[1] // lora_group_k in (2, 3))))
                 or (rp_ in WSK_RANKS and rp_ > 16 and not lora_group_k and ln is None and lora[1].shape[1] == rp_ and WSK_PACK and _wsk_is_frozen(W)))      # rank pads 32 / 64: packed weights only
            and wsk_shape(X.shape[0], W.shape[0], W.shape[1], lora is not None)):
        # 1280-wide product at batch 1 with exactly one 64 x 80 tile per CU: K split over the waves, rank-16 adapter fused (sdlt_wsk_gemm)
        _chk2(X), _chk2(W), _chk2(out)
        M_, N_, K_ = X.shape[0], W.shape[0], W.shape[1]
        assert X.shape[1] == K_ and tuple(out.shape) == (M_, N_)
        if bias is not None:
            _chk2(bias, F32)
        if residual is not None:
            _chk2(residual)
        A_ = B_ = T_ = None
        scale_ = 0.0
        if lora is not None:
            A_, B_, scale_, T_ = lora
            _chk2(A_), _chk2(B_)
            G_ = K_ // lora_group_k if lora_group_k else 1
            assert tuple(A_.shape) == (rp_, K_) and tuple(B_.shape) == (N_, rp_ * G_)
            if T_ is not None:
                _chk2(T_)
                assert tuple(T_.shape) == (M_, rp_ * G_)
        Wptr, Wld = _wsk_operand(W)
        hint = _pf_hint(Wptr, Wld, N_, K_)
        if rp_ > 16 or col_scale is not None or out0 is not None or hint is not None:
            # the parameter-block entry point: rank pads 32 / 64 (packed weights) and DoRA's column factor exist only there
            q = _lib.WskGemmParams()
            q.X, q.ldx, q.W, q.ldw, q.M, q.N, q.K = _p(X), _ld(X), Wptr, Wld, M_, N_, K_
            q.bias, q.Y, q.ldy = _p(bias), _p(out), _ld(out)
            if hint is not None:
                q.pf_next_w, q.pf_next_n, q.pf_next_k, q.pf_steps = hint[0], hint[1], hint[2], min(hint[2] // 64, WSK_PREFETCH_STEPS)
            if lora is not None:
                q.Adown, q.ld_adown, q.Bup, q.ld_bup, q.lora_scale, q.lora_rp = _p(A_), _ld(A_), _p(B_), _ld(B_), float(scale_), rp_
            if T_ is not None:
                q.T_out, q.ld_t = _p(T_), _ld(T_)
            q.lora_group_k = int(lora_group_k)
            if col_scale is not None:
                _chk2(col_scale, F32)
                assert col_scale.numel() == N_
                q.col_scale = _p(col_scale)
            if residual is not None:
                q.R, q.ldr = _p(residual), _ld(residual)
            if out0 is not None:
                _chk2(out0)
                q.Y0, q.ldy0 = _p(out0), _ld(out0)
            if ln is not None:
                c1, stats_, eps_, lnad = ln[:4]
                _chk2(c1, F32)
                assert c1.numel() == N_ and (stats_ is None or (stats_.dtype == F32 and stats_.numel() >= 2 * M_)) and (lora is None or (lnad is not None and lnad.numel() >= 2 * rp_))
                q.ln_c1, q.ln_stats, q.ln_eps, q.ln_adapter = _p(c1), _p(stats_), float(eps_), _p(lnad)
            elif ln_parts_out is not None:
                assert ln_parts_out.dtype == F32 and ln_parts_out.is_contiguous() and ln_parts_out.numel() >= M_ * (N_ // 80) * 2
                q.ln_parts = _p(ln_parts_out)
            elif rowdot is not None and ROWDOT and residual is None and N_ % 64 == 0 and M_ % rowdot["Nq"] == 0:
                O_, D_ = rowdot["O"], rowdot["D"]
                _chk2(O_), _chk2(D_, F32)
                assert tuple(O_.shape) == (M_, N_) and D_.is_contiguous() and D_.numel() >= M_ * (N_ // 64)
                q.R, q.ldr, q.dotD, q.dot_nq = _p(O_), _ld(O_), _p(D_), int(rowdot["Nq"])
                rowdot["done"] = True
            assert rp_ <= 16 or Wld == 0, "rank pad 32 runs on the packed copy of a frozen weight"
            _lib.check(lib.sdlt_wsk_gemm_p(C.byref(q), _stream()), "sdlt_wsk_gemm_p")
            return out
        if ln is not None:
            c1, stats_, eps_, lnad = ln[:4]
            _chk2(c1, F32)
            assert c1.numel() == N_ and (stats_ is None or (stats_.dtype == F32 and stats_.numel() >= 2 * M_)) and (lora is None or lnad is not None)
            _lib.check(lib.sdlt_wsk_gemm_ln(_p(X), _ld(X), Wptr, Wld, M_, N_, K_, _p(bias), _p(residual), _ld(residual) if residual is not None else 0,
                                            _p(out), _ld(out), _p(A_), _ld(A_) if A_ is not None else 0, _p(B_), _ld(B_) if B_ is not None else 0, float(scale_),
                                            _p(T_), _ld(T_) if T_ is not None else 0, _p(c1), _p(stats_), float(eps_), _p(lnad), _stream()), "sdlt_wsk_gemm_ln")
            return out
        if ln_parts_out is not None:
            assert ln_parts_out.dtype == F32 and ln_parts_out.is_contiguous() and ln_parts_out.numel() >= M_ * (N_ // 80) * 2
            _lib.check(lib.sdlt_wsk_gemm_parts(_p(X), _ld(X), Wptr, Wld, M_, N_, K_, _p(bias), _p(residual), _ld(residual) if residual is not None else 0,
                                               _p(out), _ld(out), _p(A_), _ld(A_) if A_ is not None else 0, _p(B_), _ld(B_) if B_ is not None else 0, float(scale_),
                                               _p(T_), _ld(T_) if T_ is not None else 0, int(lora_group_k) if lora is not None else 0, _p(ln_parts_out), _stream()),
                       "sdlt_wsk_gemm_parts")
            return out
        if rowdot is not None and ROWDOT and residual is None and N_ % 64 == 0 and M_ % rowdot["Nq"] == 0:
            O_, D_ = rowdot["O"], rowdot["D"]
            _chk2(O_), _chk2(D_, F32)
            assert tuple(O_.shape) == (M_, N_) and D_.is_contiguous() and D_.numel() >= M_ * (N_ // 64)
            _lib.check(lib.sdlt_wsk_gemm_rowdot(_p(X), _ld(X), Wptr, Wld, M_, N_, K_, _p(bias), _p(O_), _ld(O_), _p(out), _ld(out), _p(A_),
                                                _ld(A_) if A_ is not None else 0, _p(B_), _ld(B_) if B_ is not None else 0, float(scale_), _p(T_),
                                                _ld(T_) if T_ is not None else 0, int(lora_group_k) if lora is not None else 0, _p(D_), int(rowdot["Nq"]), _stream()),
                       "sdlt_wsk_gemm_rowdot")
            rowdot["done"] = True
            return out
        _lib.check(lib.sdlt_wsk_gemm(_p(X), _ld(X), Wptr, Wld, M_, N_, K_, _p(bias), _p(residual), _ld(residual) if residual is not None else 0,
                                     _p(out), _ld(out), _p(A_), _ld(A_) if A_ is not None else 0, _p(B_), _ld(B_) if B_ is not None else 0, float(scale_),
                                     _p(T_), _ld(T_) if T_ is not None else 0, int(lora_group_k) if lora is not None else 0, _stream()), "sdlt_wsk_gemm")
        return out
    assert ln_parts_out is None, "ln_parts_out: this product does not run on the wave-split-K kernel (ops.gemm_emits_parts)"
    if out0 is not None:          # (no wave-split-K shape: the product without the residual, then the add as its own launch)
        gemm(X, W, out0, lora=lora, bias=bias, col_scale=col_scale, tile=tile, splitk=splitk, stages=stages, lora_group_n=lora_group_n, lora_group_k=lora_group_k)
        return add2d(out0, residual, out)
    if (conv is not None and X2 is None and alpha == 1.0 and Ct is None and batch is None and geglu_out is None and geglu_bwd is None and act_out is None
            and dact_in is None and col_scale is None and not accumulate and tile == 0 and splitk == 0 and not lora_group_n and not lora_group_k and ln is None
            and out is not None and out.dtype == BF16 and not THROUGHPUT_HINT and (lora is None or (lora[0].shape[0] == 16 and lora[1].shape[1] == 16))
            and wsk_conv_shape(conv, W.shape[0], 16 if lora is not None else 0)):
        # 3 x 3 convolution of the 32 x 32 level: one 64 x 80 tile per CU, K = 9 Cin split over the waves, no split-K partials through HBM (sdlt_wsk_conv)
        _chk2(X), _chk2(W), _chk2(out)
        M_, N_ = conv.B * conv.Hout * conv.Wout, W.shape[0]
        assert W.shape[1] == 9 * conv.Cin and X.shape[1] == conv.Cin and X.shape[0] == M_ and tuple(out.shape) == (M_, N_)
        A_ = B_ = T_ = None
        scale_ = 0.0
        if lora is not None:
            A_, B_, scale_, T_ = lora
            _chk2(A_), _chk2(B_)
            assert tuple(A_.shape) == (16, 9 * conv.Cin) and tuple(B_.shape) == (N_, 16) and (T_ is None or tuple(T_.shape) == (M_, 16))
        if bias is not None:
            _chk2(bias, F32)
        if rowbias is not None:
            _chk2(rowbias)
            assert rowbias.shape[1] == N_ and rows_per_batch == conv.Hout * conv.Wout and rowbias.shape[0] >= conv.B
        if residual is not None:
            _chk2(residual)
            assert tuple(residual.shape) == (M_, N_)
        Wptr, Wld = _wsk_operand(W)
        _lib.check(lib.sdlt_wsk_conv(_p(X), _ld(X), Wptr, Wld, conv.B, conv.Hout, conv.Wout, conv.Cin, N_, int(conv.flip), _p(bias), _p(rowbias),
                                     _ld(rowbias) if rowbias is not None else 0, _p(residual), _ld(residual) if residual is not None else 0, _p(out), _ld(out),
                                     _p(A_), _ld(A_) if A_ is not None else 0, _p(B_), _ld(B_) if B_ is not None else 0, float(scale_), _p(T_),
                                     _ld(T_) if T_ is not None else 0, _p(zero_page(X.device)), _stream()), "sdlt_wsk_conv")
        return out
    p = _lib.GemmParams()
    _chk2(X), _chk2(W)
    p.X, p.ldx, p.W, p.ldw = _p(X), _ld(X), _p(W), _ld(W)
    N, K = W.shape
    if conv is None:
        M = X.shape[0]
        assert X.shape[1] == K, (X.shape, W.shape)
        p.mode = 0
    else:
        M = conv.B * conv.Hout * conv.Wout
        assert K == 9 * conv.Cin and X.shape[1] == conv.Cin and X.shape[0] == conv.B * conv.Hin * conv.Win
        p.mode = 1
        p.Hin, p.Win, p.Cin, p.Hout, p.Wout = conv.Hin, conv.Win, conv.Cin, conv.Hout, conv.Wout
        p.stride, p.ups, p.flip, p.tr = conv.stride, conv.ups, conv.flip, conv.tr
        p.zero = _p(zero_page(X.device))
    p.M, p.N, p.K = M, N, K
    p.throughput_hint = int(THROUGHPUT_HINT)
    if X2 is not None:
        _chk2(X2), _chk2(W2)
        assert X2.shape[0] == M and W2.shape[0] == N and X2.shape[1] == W2.shape[1]
        p.X2, p.ldx2, p.W2, p.ldw2, p.K2 = _p(X2), _ld(X2), _p(W2), _ld(W2), X2.shape[1]
    if lora is not None:
        Adown, Bup, scale, T_out = lora
        _chk2(Adown), _chk2(Bup)
        if lora_group_k:       # K-grouped: one adapter (padded rank Rp) per group of input columns
            assert K % lora_group_k == 0 and not lora_group_n
            G, Rp = K // lora_group_k, Adown.shape[0]
            assert Rp in (16, 32, 64) and tuple(Adown.shape) == (Rp, K) and tuple(Bup.shape) == (N, G * Rp), (Adown.shape, Bup.shape)
            p.lora_group_k = lora_group_k
        else:
            Rp, G = Bup.shape[1], 1
            if lora_group_n:   # N-grouped: one adapter per group of output columns
                assert N % lora_group_n == 0
                G = N // lora_group_n
                p.lora_group_n = lora_group_n
            assert tuple(Adown.shape) == (G * Rp, K) and Bup.shape[0] == N, (Adown.shape, Bup.shape, N, K, G)
        p.Adown, p.ld_adown, p.Bup, p.ld_bup = _p(Adown), _ld(Adown), _p(Bup), _ld(Bup)
        p.lora_R, p.lora_scale = Rp, float(scale)
        if T_out is not None:
            _chk2(T_out)
            assert tuple(T_out.shape) == (M, G * Rp)
            p.T_out, p.ld_t = _p(T_out), _ld(T_out)
    p.alpha = float(alpha)
    if col_scale is not None:
        _chk2(col_scale, F32)
        assert lora is not None and col_scale.numel() == N
        p.col_scale = _p(col_scale)
    if bias is not None:
        _chk2(bias, F32)
        assert bias.numel() == N
        p.bias = _p(bias)
    if rowbias is not None:
        _chk2(rowbias)
        assert rowbias.shape[1] == N and rows_per_batch > 0
        p.rowbias, p.ld_rowbias, p.rows_per_batch = _p(rowbias), _ld(rowbias), rows_per_batch
    if residual is not None:
        _chk2(residual)
        assert tuple(residual.shape) == (M, N)
        p.R, p.ldr = _p(residual), _ld(residual)
    if geglu_bwd is not None:
        f1, df1 = geglu_bwd
        _chk2(f1), _chk2(df1)
        assert out is None and tuple(f1.shape) == (M, 2 * N) and tuple(df1.shape) == (M, 2 * N)
        p.epi_op, p.epi_in, p.ld_epi_in, p.epi_out, p.ld_epi_out = 2, _p(f1), _ld(f1), _p(df1), _ld(df1)
    else:
        assert out.is_cuda and tuple(out.shape) == (M, N) and out.dtype in (BF16, F32)
        p.C, p.ldc, p.out_fp32 = _p(out), _ld(out), int(out.dtype == F32)
    if act_out is not None or dact_in is not None:
        assert geglu_out is None and geglu_bwd is None and not (act_out is not None and dact_in is not None)
        kind, t = act_out if act_out is not None else dact_in
        _chk2(t)
        assert tuple(t.shape) == (M, N) and kind in ("gelu", "quick_gelu")
        p.epi_act = int(kind == "quick_gelu")
        if act_out is not None:
            p.epi_op, p.epi_out, p.ld_epi_out = 3, _p(t), _ld(t)
        else:
            p.epi_op, p.epi_in, p.ld_epi_in = 4, _p(t), _ld(t)
    if geglu_out is not None:
        _chk2(geglu_out)
        assert tuple(geglu_out.shape) == (M, N // 2) and geglu_bwd is None
        p.epi_op, p.epi_out, p.ld_epi_out = 1, _p(geglu_out), _ld(geglu_out)
    if Ct is not None:
        _chk2(Ct)
        assert Ct.shape[0] == N and Ct.shape[1] >= M
        p.Ct, p.ldct = _p(Ct), _ld(Ct)
    p.tile, p.splitk, p.stages, p.accumulate = tile, splitk, stages, int(accumulate)
    if ln is not None:
        c1, stats_, eps_, lnad = ln[:4]
        _chk2(c1, F32)
        assert c1.numel() == N and (stats_ is None or (stats_.dtype == F32 and stats_.numel() >= 2 * M)) and (lora is None or lnad is not None)
        p.ln_c1, p.ln_stats, p.ln_eps, p.ln_adapter = _p(c1), _p(stats_), float(eps_), _p(lnad)
        if len(ln) > 4 and ln[4] is not None:
            parts, npart = ln[4], int(ln[5])
            assert parts.dtype == F32 and parts.is_contiguous() and parts.numel() >= M * npart * 2
            p.ln_parts, p.ln_nparts = _p(parts), npart
    if batch is not None:
        p.batch, p.n_batch = _p(batch.dev), batch.n
    slab, cnt = splitk_workspace(X.device)
    p.ws_slab, p.ws_slab_bytes, p.ws_cnt, p.ws_cnt_len = _p(slab), slab.numel(), _p(cnt), cnt.numel()
    _lib.check(lib.sdlt_gemm_bf16(C.byref(p), _stream()), "sdlt_gemm_bf16")
    return out if geglu_bwd is None else geglu_bwd[1]


STRIP_SPLITK = int(os.environ.get("SDLT_STRIP_SPLITK", "1"))      # in-kernel K split of strip_gemm: 1 = off (default), n = force where K allows


def strip_splitk(N, K, B):
    """In-kernel K split (last-arriver reduction) of a row-strip product.  Measured (tools/strip_probe.py): the seam - write-through slab
    stores, drain, ticket, acquire, slab reads - costs 4 - 5 us, more than the shorter K walk saves at every CLIP shape (1280 x 5120: 13.3 us
    unsplit, 17.7 us with 4 slices; 1280 x 3840: 10.2 vs 10.0), so it stays off; the long-K products of the backward pass are split with
    their reduction in the consumer's prologue instead (strip_partial_splits)."""
    s = max(1, STRIP_SPLITK)
    return s if (K // 256) >= s else 1


def strip_partial_splits(N, K, B):
    """Splits of a row-strip product whose partial tiles are added by its consumer (strip_gemm(partial=...) -> layernorm_bwd(dy_slabs=...)):
    as many as keep the grid within ONE wave of workgroups (a strip workgroup takes a CU's whole LDS) with at least 3 K steps each."""
    strips = N // 16
    s = max(1, min(8, 256 // max(1, strips * B), (K // 256) // 3))
    return s


# ---- paired launches (sdlt_*_pair): two independent chains of the same kernels - the two text encoders of SDXL - issued in lockstep.
# Inside `with pairing() as pq:` the four pairable ops below do not launch: they leave ONE pending (kind, params) record per call in pq;
# the driver (step.TextStack) advances chain a by one op, chain b by one op, then calls pq.flush(), which launches the two records as one
# paired launch when the C side takes the combination and as two single launches otherwise.
STRIP_WIDE_MIN = int(os.environ.get("SDLT_STRIP_WIDE_MIN", "2304"))


class _PairQueue:
    def __init__(self):
        self.pending = []
        self.launches = 0

    def take(self):
        """The records left since the last take() (the ops of the chain that was just advanced)."""
        rec, self.pending = self.pending, []
        return rec

    def launch(self, recs):
        """One paired launch when `recs` is two records of one kind the C side takes together, else one single launch per record."""
        lib, st = _lib.load(), _stream()
        if len(recs) == 2 and recs[0][0] == recs[1][0] and _pair_ok(lib, recs[0], recs[1]):
            kind = recs[0][0]
            _lib.check(getattr(lib, _PAIR_FN[kind])(C.byref(recs[0][1]), C.byref(recs[1][1]), st), _PAIR_FN[kind])
            self.launches += 1
            return
        for kind, p, _ in recs:
            _launch_single(lib, kind, p, st)
            self.launches += 1

    def flush(self):
        self.launch(self.take())


def run_paired(gens, prefer):
    """Drive two generators (each yields a key after every pairable op it issued, other work runs inline) in lockstep: ops with equal keys
    go out as one paired launch; with unequal keys the chain `prefer(key_a, key_b)` names goes first alone.  Returns the generators' values."""
    vals, keys, pend, done = [None, None], [None, None], [None, None], [False, False]
    with pairing() as pq:
        while True:
            for i in (0, 1):
                if not done[i] and pend[i] is None:
                    try:
                        keys[i] = next(gens[i])
                    except StopIteration as e:
                        vals[i], done[i] = e.value, True
                    rec = pq.take()
                    if len(rec) > 1 or (done[i] and rec):      # a chain must yield after every pairable op
                        pq.launch(rec)
                        raise AssertionError("run_paired: a chain issued a pairable op without yielding")
                    pend[i] = rec[0] if rec else None
            if pend[0] is not None and pend[1] is not None:
                if keys[0] == keys[1]:
                    pq.launch([pend[0], pend[1]])
                    pend = [None, None]
                else:
                    i = 0 if prefer(keys[0], keys[1]) == keys[0] else 1
                    pq.launch([pend[i]])
                    pend[i] = None
            elif pend[0] is not None or pend[1] is not None:
                i = 0 if pend[0] is not None else 1
                pq.launch([pend[i]])
                pend[i] = None
            elif done[0] and done[1]:
                return vals


_PAIR_FN = {"strip": "sdlt_strip_gemm_pair", "attn_fwd": "sdlt_attn_fwd_pair", "attn_bwd": "sdlt_attn_bwd_pair", "ln_slabs": "sdlt_layernorm_bwd_slabs_pair"}
_pair_queue = None


class pairing:
    def __enter__(self):
        global _pair_queue
        assert _pair_queue is None
        _pair_queue = _PairQueue()
        return _pair_queue

    def __exit__(self, *exc):
        global _pair_queue
        q, _pair_queue = _pair_queue, None
        if exc[0] is None:
            q.flush()
        return False


def _pair_ok(lib, a, b):
    kind, pa, pb = a[0], a[1], b[1]
    if kind == "strip":
        wide = lambda n: n >= STRIP_WIDE_MIN and n % 32 == 0  # noqa: E731  (strip.hip strip_wide)
        return bool(pa.ln) == bool(pb.ln) and wide(pa.N) == wide(pb.N) and not (pa.splitk > 1 and not pa.P) and not (pb.splitk > 1 and not pb.P)
    if kind in ("attn_fwd", "attn_bwd"):
        return bool(lib.sdlt_attn_pair_ok(C.byref(pa), C.byref(pb)))
    return True


def _launch_single(lib, kind, p, st):
    if kind == "strip":
        _lib.check(lib.sdlt_strip_gemm(C.byref(p), st), "sdlt_strip_gemm")
    elif kind == "attn_fwd":
        _lib.check(lib.sdlt_attn_fwd(C.byref(p), st), "sdlt_attn_fwd")
    elif kind == "attn_bwd":
        _lib.check(lib.sdlt_attn_bwd(C.byref(p), st), "sdlt_attn_bwd")
    else:
        _lib.check(lib.sdlt_layernorm_bwd_slabs(p.x, p.ldx, p.dy32, p.lddy32, p.nslab, p.M, p.C, p.gamma, p.stats, p.dres, p.lddres, p.dx, p.lddx, st),
                   "sdlt_layernorm_bwd_slabs")


def _issue(kind, p, keep=None):
    """Launch now, or leave the record with the active pairing queue (keep: tensors the record's pointers refer to)."""
    if _pair_queue is not None:
        _pair_queue.pending.append((kind, p, keep))
    else:
        _launch_single(_lib.load(), kind, p, _stream())


def strip_gemm(X, W, out, *, B, T, Tp, bias=None, residual=None, act_out=None, dact_in=None, ln=None, stats=None, splitk=None, partial=None):
    """Row-strip product of the text encoders (sdlt_strip_gemm): out[b*Tp + t] = X[b*Tp + t] . W^T for the t < T valid rows of every batch
    element; rows t >= T of `out` are left untouched.  W [N, K], K % 256 == 0.
    ln = (c1 [N], c2 [N], eps): a LayerNorm sits in front of the product - X holds the raw rows, W = (weight o gamma), c1 = rowsum(W),
    c2 = weight . beta + bias (fold_layernorm); stats [B*Tp, 2] receives (mean, rstd) for layernorm_bwd.
    act_out = (kind, A): also A = act(out);  dact_in = (kind, P): out = (product + bias + residual) * act'(P)."""
    lib = _lib.load()
    p = _lib.StripParams()
    _chk2(X), _chk2(W)
    N, K = W.shape
    assert X.shape[0] == B * Tp and X.shape[1] == K, (X.shape, W.shape)
    p.X, p.ldx, p.W, p.ldw = _p(X), _ld(X), _p(W), _ld(W)
    p.B, p.T, p.Tp, p.N, p.K = B, T, Tp, N, K
    if partial is not None:
        # partial fp32 [S, B*Tp, N]: the K split's tiles, added by the consumer (layernorm_bwd(dy_slabs=partial)); `out` is not written
        assert out is None and bias is None and residual is None and act_out is None and dact_in is None and ln is None
        assert partial.is_cuda and partial.dtype == F32 and partial.is_contiguous() and tuple(partial.shape[1:]) == (B * Tp, N)
        p.P, p.ldp, p.splitk = _p(partial), N, partial.shape[0]
        _issue("strip", p)
        return partial
    _chk2(out)
    assert tuple(out.shape) == (B * Tp, N), (out.shape, W.shape)
    p.Y, p.ldy = _p(out), _ld(out)
    if ln is not None:
        c1, c2, eps = ln
        _chk2(c1, F32), _chk2(c2, F32)
        assert c1.numel() == N and c2.numel() == N and bias is None, "with ln the bias is part of c2"
        p.ln, p.c1, p.c2, p.eps = 1, _p(c1), _p(c2), float(eps)
        if stats is not None:
            _chk2(stats, F32)
            assert stats.numel() >= B * Tp * 2
            p.stats = _p(stats)
    elif bias is not None:
        _chk2(bias, F32)
        assert bias.numel() == N
        p.bias = _p(bias)
    if residual is not None:
        _chk2(residual)
        assert tuple(residual.shape) == (B * Tp, N)
        p.R, p.ldr = _p(residual), _ld(residual)
    if act_out is not None or dact_in is not None:
        assert not (act_out is not None and dact_in is not None)
        kind, t = act_out if act_out is not None else dact_in
        _chk2(t)
        assert tuple(t.shape) == (B * Tp, N) and kind in ("gelu", "quick_gelu")
        p.act = 1 if kind == "quick_gelu" else 2
        if act_out is not None:
            p.Y2, p.ldy2 = _p(t), _ld(t)
        else:
            p.Z, p.ldz = _p(t), _ld(t)
    S = strip_splitk(N, K, B) if splitk is None else splitk
    if S > 1:
        slab, cnt = splitk_workspace(X.device)
        p.splitk, p.ws, p.ws_bytes, p.cnt, p.cnt_len = S, _p(slab), slab.numel(), _p(cnt), cnt.numel()
    _issue("strip", p)
    return out


class TokenAttentionPlan:
    """sdlt_token_attention_loss on persistent buffers: groups = [(S fp32 [B*h*w, 128], h, w, Wh [h0,h] | None, Ww [w0,w] | None)] sorted by size
    (smallest first); owns the gradient operands dS / dSt per group, the scratch and the loss scalar."""

    def __init__(self, groups, B, n_tok, n_layers, mask, tok_w, tok_cnt, ti_onehot, has_ti, loss, act_dtype=BF16):
        dev = mask.device
        p = _lib.TaParams()
        self.keep, self.out = [groups, mask, tok_w, tok_cnt, ti_onehot, has_ti, loss], {}
        for gi, (S, h, w, Wh, Ww) in enumerate(groups):
            N = h * w
            _chk2(S, F32)
            assert S.shape == (B * N, 128)
            dS, dSt = torch.zeros(B * N, 128, dtype=act_dtype, device=dev), torch.zeros(B * 128, N, dtype=act_dtype, device=dev)
            dheat = torch.zeros(B * n_tok * N, dtype=F32, device=dev)
            g = p.g[gi]
            g.S, g.dS, g.dSt, g.dheat, g.h, g.w = _p(S), _p(dS), _p(dSt), _p(dheat), h, w
            if gi > 0:
                Wh, Ww = Wh.to(dev, F32).contiguous(), Ww.to(dev, F32).contiguous()
                ch, cw = Wh.sum(0).contiguous(), Ww.sum(0).contiguous()
                g.Wh, g.Ww, g.ch, g.cw = _p(Wh), _p(Ww), _p(ch), _p(cw)
                self.keep += [Wh, Ww, ch, cw]
            self.keep += [dS, dSt, dheat]
            self.out[N] = (dS, dSt)
        for t in (mask, tok_w, tok_cnt, ti_onehot, has_ti, loss):
            assert t.is_cuda and t.dtype == F32 and t.is_contiguous()
        p.mask, p.tok_w, p.tok_cnt, p.ti_onehot, p.has_ti, p.loss = _p(mask), _p(tok_w), _p(tok_cnt), _p(ti_onehot), _p(has_ti), _p(loss)
        p.ngroups, p.B, p.n_tok, p.n_layers, p.mH, p.mW = len(groups), B, n_tok, n_layers, mask.shape[2], mask.shape[3]
        nws = _lib.load().sdlt_token_attention_ws_floats(C.byref(p))
        self.ws = torch.zeros(nws, dtype=F32, device=dev)
        p.ws, p.ws_floats = _p(self.ws), nws
        self.p = p

    def run(self, weight):
        self.p.weight = float(weight)
        _lib.check(_lib.load().sdlt_token_attention_loss(C.byref(self.p), _stream()), "sdlt_token_attention_loss")
        return self.out


def fold_layernorm(W, bias, gamma, beta, dtype=BF16):
    """Operands of a LayerNorm folded into the Linear behind it (strip_gemm ln=): LN(x) W^T + b = rstd (x (W o gamma)^T - mean c1) + c2
    with c1[n] = sum_k (W o gamma)[n,k] taken from the ROUNDED operand (so that the mean term cancels exactly what the product adds)
    and c2 = W beta + b.  W fp32 [N, K].  Returns (W o gamma in `dtype`, c1 fp32, c2 fp32)."""
    Wg = (W.float() * gamma.float()[None, :]).to(dtype).contiguous()
    c1 = Wg.float().sum(1).contiguous()
    c2 = (W.float() @ beta.float()).contiguous()
    if bias is not None:
        c2 = (c2 + bias.float()).contiguous()
    return Wg, c1, c2


def geglu_perm(H, device=None):
    """Row permutation of ff.net.0.proj's weight ([2H, K]: hidden rows then gate rows) into the interleaved-16 layout of the fused
    GEGLU epilogues: new row (j // 16) * 32 + j % 16 = hidden j, 16 rows further its gate."""
    j = torch.arange(H, device=device)
    pos_h = (j // 16) * 32 + j % 16
    perm = torch.empty(2 * H, dtype=torch.int64, device=device)
    perm[pos_h] = j
    perm[pos_h + 16] = H + j
    return perm


class GemmBatch:
    """Device-resident operand table of a batched sdlt_gemm_bf16 launch (sdlt_gemm_batch_item[]).
    items: list of dict with tensors (or None) under X, W, Adown, Bup, T_out, C, Ct, bias - same shapes and strides per key."""
    KEYS = ("X", "W", "Adown", "Bup", "T_out", "C", "Ct", "bias", "col_scale")

    def __init__(self, items, device):
        arr = (_lib.GemmBatchItem * len(items))()
        self.keep = []
        for i, it in enumerate(items):
            for k in self.KEYS:
                t = it.get(k)
                if t is not None:
                    assert t.is_cuda
                    ref = items[0].get(k)
                    assert ref is not None and t.shape == ref.shape and t.stride() == ref.stride() and t.dtype == ref.dtype, f"batch item {i}: {k} differs from item 0"
                    setattr(arr[i], k, t.data_ptr())
                    self.keep.append(t)
        self.n = len(items)
        self.items = items
        self.dev = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(device)


class LoraGradPlan:
    """Device-resident descriptor table for sdlt_lora_grad_grouped (built once, replayed every step)."""

    def __init__(self, problems, Rp, device):
        # problems: list of dict(P, Q, out, M, Cw, R, rank_major, conv=None|ConvGeom)
        lib = _lib.load()
        bc = lib.sdlt_lora_grad_block_cols()
        # longest token streams first: a workgroup streams all M rows of its 64 columns, so the tail of the launch
        # should be made of the short problems
        order = sorted(range(len(problems)), key=lambda i: -problems[i]["M"])
        descs = (_lib.LoraGradDesc * len(problems))()
        block_desc = []
        self.keep = []
        nb = 0
        mfma = True
        zp = zero_page(device)
        for i, pi in enumerate(order):
            pr = problems[pi]
            d = descs[i]
            P, Q, out = pr["P"], pr["Q"], pr["out"]
            _chk2(P), _chk2(Q), _chk2(out, F32)
            d.P, d.ldp, d.Q, d.ldq, d.out = P.data_ptr(), _ld(P), Q.data_ptr(), _ld(Q), out.data_ptr()
            d.M, d.Cw, d.R, d.Rp = pr["M"], pr["Cw"], pr["R"], Rp
            assert Q.shape[1] == Rp and Q.shape[0] == pr["M"] and out.numel() == pr["Cw"] * pr["R"] and out.is_contiguous()
            d.rank_major, d.accumulate = int(pr["rank_major"]), 0
            d.zero = zp.data_ptr()
            cv = pr.get("conv")
            ok = _ld(P) % 8 == 0 and _ld(Q) % 8 == 0 and pr["Cw"] % 8 == 0 and P.data_ptr() % 16 == 0 and Q.data_ptr() % 16 == 0
            if cv is not None:
                d.conv, d.Hin, d.Win, d.Cin, d.Hout, d.Wout, d.stride = 1, cv.Hin, cv.Win, cv.Cin, cv.Hout, cv.Wout, cv.stride
                assert pr["Cw"] == 9 * cv.Cin and pr["M"] == cv.B * cv.Hout * cv.Wout
                ok = ok and cv.Cin % bc == 0
            else:
                assert P.shape[0] == pr["M"] and P.shape[1] == pr["Cw"]
            mfma = mfma and ok
            d.first_block = nb
            blocks = (pr["Cw"] + bc - 1) // bc
            block_desc += [i] * blocks
            nb += blocks
            self.keep += [P, Q, out]
        self.n_blocks, self.Rp, self.mfma = nb, Rp, int(mfma)
        raw = bytes(descs)
        self.descs_dev = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(device)
        self.block_desc_dev = torch.tensor(block_desc, dtype=torch.int32, device=device)
        self._accum_off = _lib.LoraGradDesc.accumulate.offset
        self._stride = C.sizeof(_lib.LoraGradDesc)

    def set_accumulate(self, flag):
        v = self.descs_dev.view(-1, self._stride)[:, self._accum_off:self._accum_off + 4]
        v.copy_(torch.tensor([int(flag), 0, 0, 0], dtype=torch.uint8, device=v.device).expand_as(v))

    def run(self):
        lib = _lib.load()
        _lib.check(lib.sdlt_lora_grad_grouped(_p(self.descs_dev), _p(self.block_desc_dev), self.n_blocks, self.Rp, self.mfma, _stream()),
                   "sdlt_lora_grad_grouped")


def _attn_params(Q, K, V, *, B, H, Nq, Nk, Nqp, Nkp, d, scale, causal=False):
    p = _lib.AttnParams()
    for t in (Q, K, V):
        _chk2(t)
    p.Q, p.ldq, p.K, p.ldk, p.V, p.ldv = _p(Q), _ld(Q), _p(K), _ld(K), _p(V), _ld(V)
    p.B, p.H, p.Nq, p.Nk, p.Nqp, p.Nkp, p.d = B, H, Nq, Nk, Nqp, Nkp, d
    p.scale, p.qsplit, p.causal = float(scale), 1, int(causal)
    return p


def attn_fwd(Q, K, V, Vt, O, L, *, B, H, Nq, Nk, Nqp, Nkp, d, scale, causal=False, zero_D=None):
    """zero_D (fp32 [B * H * Nq], optional): cleared by the launch - the slots gemm(rowdot=) accumulates the backward's row term into."""
    lib = _lib.load()
    p = _attn_params(Q, K, V, B=B, H=H, Nq=Nq, Nk=Nk, Nqp=Nqp, Nkp=Nkp, d=d, scale=scale, causal=causal)
    _chk2(O), _chk2(L, F32)      # Vt is accepted for call compatibility and ignored: the kernel transposes V tiles in LDS
    p.O, p.ldo, p.L = _p(O), _ld(O), _p(L)
    if zero_D is not None:
        _chk2(zero_D, F32)
        assert zero_D.is_contiguous() and zero_D.numel() >= B * H * Nq
        p.D = _p(zero_D)
    _issue("attn_fwd", p)
    return O


def attn_bwd(Q, K, V, Kt, Qt, O, L, dO, dOt, D, dQ, dK, dV, *, B, H, Nq, Nk, Nqp, Nkp, d, scale, causal=False,
             qsplit=1, dK32=None, dV32=None, accumulate_dq=False, accumulate_dk=False, defer_splitsum=False, d_ready=False):
    """defer_splitsum (single-pass cross-attention backward): dK / dV stay as per-split fp32 slabs in dK32 / dV32 (layer-owned buffers);
    a SplitsumPlan over all layers sums them in one launch later.
    d_ready (self-attention): D already holds rowsum(dO o O) - the GEMM that produced dO left it (gemm(rowdot=)) - and the pre-pass launch is skipped."""
    lib = _lib.load()
    p = _attn_params(Q, K, V, B=B, H=H, Nq=Nq, Nk=Nk, Nqp=Nqp, Nkp=Nkp, d=d, scale=scale, causal=causal)
    p.defer_splitsum = int(defer_splitsum)
    p.d_ready = int(d_ready)
    for t in (O, dO, dQ, dK, dV):      # Kt / Qt / dOt: accepted and ignored (see attn_fwd)
        _chk2(t)
    _chk2(L, F32), _chk2(D, F32)
    p.O, p.ldo, p.L, p.dO, p.lddo, p.D = _p(O), _ld(O), _p(L), _p(dO), _ld(dO), _p(D)
    p.dQ, p.lddq, p.dK, p.lddk, p.dV, p.lddv = _p(dQ), _ld(dQ), _p(dK), _ld(dK), _p(dV), _ld(dV)
    p.qsplit = qsplit
    p.accumulate_dq, p.accumulate_dk = int(accumulate_dq), int(accumulate_dk)
    if qsplit > 1:
        _chk2(dK32, F32), _chk2(dV32, F32)
        if not causal and Nkp <= 128 and d <= 96:   # single-pass cross-attention kernel: one partial slab per query split
            assert dK32.shape[0] >= qsplit * B * Nkp and dV32.shape[0] >= qsplit * B * Nkp, "dK32/dV32 must hold qsplit slabs"
        p.dK32, p.dV32, p.ld32 = _p(dK32), _p(dV32), _ld(dK32)
    _issue("attn_bwd", p)


class SplitsumPlan:
    """One launch for the partial dK / dV slabs of all cross-attention layers (sdlt_attn_splitsum_batch).
    items: dict(dK32, dV32 fp32 [nsplit*B*Nkp, C], dK, dV bf16 [B*Nkp, C] (strided ok), nsplit, B, Nk, Nkp, acc0)."""

    def __init__(self, items, device):
        arr = (_lib.SplitsumDesc * len(items))()
        counts = []
        for d, it in zip(arr, items):
            s0, s1, o0, o1 = it["dK32"], it["dV32"], it["dK"], it["dV"]
            _chk2(s0, F32), _chk2(s1, F32), _chk2(o0), _chk2(o1)
            Cw = o0.shape[1]
            assert Cw % 4 == 0 and _ld(s0) == _ld(s1) and s0.shape[0] >= it["nsplit"] * it["B"] * it["Nkp"]
            d.s0, d.s1, d.ld32, d.out0, d.ldo0, d.out1, d.ldo1 = _p(s0), _p(s1), _ld(s0), _p(o0), _ld(o0), _p(o1), _ld(o1)
            d.nsplit, d.B, d.Nk, d.Nkp, d.C, d.acc0 = it["nsplit"], it["B"], it["Nk"], it["Nkp"], Cw, int(bool(it.get("acc0")))
            d.nblocks = max(1, min(64, (it["B"] * it["Nkp"] * Cw // 2 + 255) // 256))
            counts.append(d.nblocks)
        self.keep = items
        self.descs_dev = _to_dev(arr, device)
        self.n_blocks, self.bd, self.bf = _block_table(counts, device)

    def run(self):
        _lib.check(_lib.load().sdlt_attn_splitsum_batch(_p(self.descs_dev), _p(self.bd), _p(self.bf), self.n_blocks, _stream()), "sdlt_attn_splitsum_batch")


_GN_WS = {}


def gn_workspace(device):
    """Scratch of the deterministic GroupNorm reductions (per-block partial sums + one counter per batch element), one per
    (device, workspace owner): only live inside a call, but two jobs replaying their graphs concurrently must not share it."""
    owner = _ws_owner if _ws_owner is not None else _gn_owner
    key = (device, ("owner", owner) if owner is not None else torch.cuda.current_stream(device).cuda_stream)
    ws = _GN_WS.get(key)
    if ws is None:
        ws = (torch.empty(1 << 20, dtype=F32, device=device), torch.zeros(256, dtype=torch.int32, device=device))
        _GN_WS[key] = ws
    return ws


def _gn_params(x1, x2, B, HW, gamma, beta, eps, silu, stats):
    p = _lib.GroupNormParams()
    ws, cnt = gn_workspace(x1.device)
    p.ws, p.ws_floats, p.cnt, p.cnt_len = _p(ws), ws.numel(), _p(cnt), cnt.numel()
    _chk2(x1), _chk2(gamma, F32), _chk2(beta, F32), _chk2(stats, F32)
    C1 = x1.shape[1]
    Ctot = C1 + (x2.shape[1] if x2 is not None else 0)
    assert gamma.numel() == Ctot and x1.shape[0] == B * HW and stats.numel() >= B * 64
    p.x1, p.ldx1, p.C1 = _p(x1), _ld(x1), C1
    if x2 is not None:
        _chk2(x2)
        p.x2, p.ldx2 = _p(x2), _ld(x2)
    p.B, p.HW, p.C = B, HW, Ctot
    p.gamma, p.beta, p.eps, p.silu, p.stats = _p(gamma), _p(beta), float(eps), int(silu), _p(stats)
    return p


def groupnorm_fwd(x1, x2, y, stats, *, B, HW, gamma, beta, eps, silu, stats_zeroed=False):
    lib = _lib.load()
    p = _gn_params(x1, x2, B, HW, gamma, beta, eps, silu, stats)      # (stats_zeroed: kept for callers; the reduction overwrites)
    _chk2(y)
    p.y, p.ldy = _p(y), _ld(y)
    _lib.check(lib.sdlt_groupnorm_fwd(C.byref(p), _stream()), "sdlt_groupnorm_fwd")
    return y


def groupnorm_colsum_splits(B, HW, Cc):
    """Row splits of the GroupNorm kernels' grid = the number of partial rows sdlt_groupnorm_bwd leaves in `colsum_ws` (fp32 [splits][B][C])."""
    return int(_lib.load().sdlt_groupnorm_ws_floats(B, HW, Cc)) // (B * Cc)


def groupnorm_bwd(x1, x2, dy, dx, stats, bstats, *, B, HW, gamma, beta, eps, silu, dres=None, stats_zeroed=False, colsum_ws=None):
    """colsum_ws (fp32, groupnorm_colsum_splits(...) * B * C floats, optional): partial column sums of dx per image, finished by ColsumFinishPlan.run()."""
    lib = _lib.load()
    p = _gn_params(x1, x2, B, HW, gamma, beta, eps, silu, stats)
    _chk2(dy), _chk2(dx), _chk2(bstats, F32)
    p.dy, p.lddy, p.dx, p.lddx, p.bstats = _p(dy), _ld(dy), _p(dx), _ld(dx), _p(bstats)
    if colsum_ws is not None:
        _chk2(colsum_ws, F32)
        assert colsum_ws.is_contiguous() and colsum_ws.numel() >= groupnorm_colsum_splits(B, HW, p.C) * B * p.C
        p.colsum_ws = _p(colsum_ws)
    if dres is not None:
        _chk2(dres)
        p.dres, p.lddres = _p(dres), _ld(dres)
    _lib.check(lib.sdlt_groupnorm_bwd(C.byref(p), _stream()), "sdlt_groupnorm_bwd")
    return dx


def groupnorm_affine_grad(x1, x2, dy, stats, dgamma, dbeta, *, B, HW, gamma, beta, eps, silu, accumulate=False):
    """d gamma / d beta of GroupNorm(32)(+SiLU) for the full fine-tune (sdlt_groupnorm_affine_grad); fp32 [C] outputs."""
    lib = _lib.load()
    p = _gn_params(x1, x2, B, HW, gamma, beta, eps, silu, stats)
    _chk2(dy), _chk2(dgamma, F32), _chk2(dbeta, F32)
    assert dgamma.numel() == p.C and dbeta.numel() == p.C and dgamma.is_contiguous() and dbeta.is_contiguous()
    p.dy, p.lddy = _p(dy), _ld(dy)
    _lib.check(lib.sdlt_groupnorm_affine_grad(C.byref(p), _p(dgamma), _p(dbeta), int(accumulate), _stream()), "sdlt_groupnorm_affine_grad")


def layernorm_affine_grad(x, dy, stats, dgamma, dbeta, accumulate=False):
    lib = _lib.load()
    _chk2(x), _chk2(dy), _chk2(stats, F32), _chk2(dgamma, F32), _chk2(dbeta, F32)
    M, Cc = x.shape
    assert dgamma.numel() == Cc and dbeta.numel() == Cc and dgamma.is_contiguous() and dbeta.is_contiguous()
    _lib.check(lib.sdlt_layernorm_affine_grad(_p(x), _ld(x), _p(dy), _ld(dy), M, Cc, _p(stats), _p(dgamma), _p(dbeta), int(accumulate), _stream()),
               "sdlt_layernorm_affine_grad")


def wgrad_transpose(x, out, colsum_acc=None):
    """out [C, Mp] <- x[M, C]^T, columns M..Mp-1 zero (Mp = out.shape[1], multiple of 64): token-contiguous GEMM panel.
    colsum_acc (fp32 [C], optional) += column sums of x (accumulated: the caller zeroes it)."""
    lib = _lib.load()
    _chk2(x), _chk2(out)
    M, Cc = x.shape
    assert out.shape[0] == Cc
    if colsum_acc is not None:
        _chk2(colsum_acc, F32)
        assert colsum_acc.numel() == Cc and colsum_acc.is_contiguous()
    _lib.check(lib.sdlt_wgrad_transpose(_p(x), _ld(x), M, Cc, _p(out), _ld(out), out.shape[1], _p(colsum_acc), _stream()), "sdlt_wgrad_transpose")
    return out


class AffineGradBatch:
    """Device table of sdlt_affine_grad_item: the d gamma / d beta of all norm layers of one shape in one launch.
    items: dict(x1, x2|None, dy, stats, gamma, beta, dgamma, dbeta); shared: groupnorm, B, HW, eps, silu."""

    def __init__(self, items, device, *, groupnorm, B, HW, eps=0.0, silu=False):
        import struct
        i0 = items[0]
        for it in items:
            for k in ("x1", "dy"):
                _chk2(it[k])
                assert it[k].shape == i0[k].shape and it[k].stride() == i0[k].stride()
            assert (it["x2"] is None) == (i0["x2"] is None)
            for k in ("stats", "dgamma", "dbeta"):
                _chk2(it[k], F32)
        ptr = lambda t: t.data_ptr() if t is not None else 0  # noqa: E731
        raw = b"".join(struct.pack("8Q", ptr(it["x1"]), ptr(it["x2"]), ptr(it["dy"]), ptr(it["stats"]), ptr(it.get("gamma")), ptr(it.get("beta")),
                                   ptr(it["dgamma"]), ptr(it["dbeta"])) for it in items)
        self.dev = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(device)
        self.items, self.n = items, len(items)
        C1 = i0["x1"].shape[1]
        self.C = C1 + (i0["x2"].shape[1] if i0["x2"] is not None else 0)
        self.args = (int(groupnorm), _ld(i0["x1"]), C1 if i0["x2"] is not None else 0, _ld(i0["x2"]) if i0["x2"] is not None else 0, _ld(i0["dy"]),
                     B, HW, self.C, float(eps), int(silu))

    def run(self):
        lib = _lib.load()
        _lib.check(lib.sdlt_affine_grad_batch(_p(self.dev), self.n, *self.args, _stream()), "sdlt_affine_grad_batch")


class WgradPanelBatch:
    """Device table of sdlt_wgrad_tr_item for the batched panel launches: items = [(x, out, colsum or None)], identical
    shapes / strides; conv = None (plain transpose) or dict(B, H, W, stride, ups) (transposed im2col)."""

    def __init__(self, items, device, conv=None):
        import struct
        x0, o0, _ = items[0]
        for x, o, cs in items:
            _chk2(x), _chk2(o)
            assert x.shape == x0.shape and x.stride() == x0.stride() and o.shape == o0.shape and o.stride() == o0.stride()
            if cs is not None:
                _chk2(cs, F32)
                assert cs.is_contiguous() and cs.numel() == x.shape[1]
        raw = b"".join(struct.pack("QQQ", x.data_ptr(), o.data_ptr(), cs.data_ptr() if cs is not None else 0) for x, o, cs in items)
        self.dev = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(device)
        self.items, self.n, self.conv = items, len(items), conv

    def run(self):
        lib = _lib.load()
        x0, o0, _ = self.items[0]
        if self.conv is None:
            _lib.check(lib.sdlt_wgrad_transpose_batch(_p(self.dev), self.n, _ld(x0), x0.shape[0], x0.shape[1], _ld(o0), o0.shape[1], _stream()),
                       "sdlt_wgrad_transpose_batch")
        else:
            c = self.conv
            _lib.check(lib.sdlt_wgrad_im2col_t_batch(_p(self.dev), self.n, _ld(x0), c["B"], c["H"], c["W"], x0.shape[1], c["stride"], c["ups"],
                                                     _ld(o0), o0.shape[1], _stream()), "sdlt_wgrad_im2col_t_batch")


def wgrad_im2col_t(x, out, *, B, H, W, stride=1, ups=1):
    """out [9*C, Mp] <- transposed im2col of the NHWC activation x [B*H*W, C] for a 3x3 pad-1 conv (rows tap*C + c)."""
    lib = _lib.load()
    _chk2(x), _chk2(out)
    Cc = x.shape[1]
    assert x.shape[0] == B * H * W and out.shape[0] == 9 * Cc
    _lib.check(lib.sdlt_wgrad_im2col_t(_p(x), _ld(x), B, H, W, Cc, stride, ups, _p(out), _ld(out), out.shape[1], _stream()),
               "sdlt_wgrad_im2col_t")
    return out


def layernorm_fwd(x, y, stats, *, gamma, beta, eps=1e-5):
    lib = _lib.load()
    _chk2(x), _chk2(y), _chk2(stats, F32), _chk2(gamma, F32), _chk2(beta, F32)
    M, Cc = x.shape
    _lib.check(lib.sdlt_layernorm_fwd(_p(x), _ld(x), M, Cc, _p(gamma), _p(beta), float(eps), _p(y), _ld(y), _p(stats), _stream()),
               "sdlt_layernorm_fwd")
    return y


def layernorm_bwd(x, dy, dx, stats, *, gamma, dres=None, dy_slabs=None, beta=None, y_out=None):
    """dy_slabs fp32 [S, M, C] (instead of dy): the partial outputs of a K-split strip_gemm, added in slab order in the kernel's prologue.
    y_out (with beta): also writes the normalised rows - the backward of a LayerNorm whose forward was folded into its consumer GEMM."""
    lib = _lib.load()
    M, Cc = x.shape
    if y_out is not None:
        assert dy_slabs is None and beta is not None
        _chk2(x), _chk2(dy), _chk2(dx), _chk2(stats, F32), _chk2(gamma, F32), _chk2(beta, F32), _chk2(y_out)
        _lib.check(lib.sdlt_layernorm_bwd_y(_p(x), _ld(x), _p(dy), _ld(dy), M, Cc, _p(gamma), _p(beta), _p(stats), _p(dres),
                                            _ld(dres) if dres is not None else 0, _p(dx), _ld(dx), _p(y_out), _ld(y_out), _stream()), "sdlt_layernorm_bwd_y")
        return dx
    if dy_slabs is not None:
        assert dy is None and dy_slabs.dtype == F32 and dy_slabs.is_contiguous() and tuple(dy_slabs.shape[1:]) == (M, Cc)
        _chk2(x), _chk2(dx), _chk2(stats, F32), _chk2(gamma, F32)
        q = _lib.LnSlabsParams()
        q.x, q.ldx, q.dy32, q.lddy32, q.nslab, q.M, q.C = _p(x), _ld(x), _p(dy_slabs), Cc, dy_slabs.shape[0], M, Cc
        q.gamma, q.stats, q.dres, q.lddres, q.dx, q.lddx = _p(gamma), _p(stats), _p(dres), _ld(dres) if dres is not None else 0, _p(dx), _ld(dx)
        _issue("ln_slabs", q)
        return dx
    _chk2(x), _chk2(dy), _chk2(dx), _chk2(stats, F32), _chk2(gamma, F32)
    _lib.check(lib.sdlt_layernorm_bwd(_p(x), _ld(x), _p(dy), _ld(dy), M, Cc, _p(gamma), _p(stats), _p(dres),
                                      _ld(dres) if dres is not None else 0, _p(dx), _ld(dx), _stream()), "sdlt_layernorm_bwd")
    return dx


def geglu_fwd(inp, out):
    lib = _lib.load()
    _chk2(inp), _chk2(out)
    M, C2 = inp.shape
    _lib.check(lib.sdlt_geglu_fwd(_p(inp), _ld(inp), M, C2 // 2, _p(out), _ld(out), _stream()), "sdlt_geglu_fwd")
    return out


def geglu_bwd(inp, dout, din):
    """(halves layout [hidden | gate]; the interleaved-16 layout only exists inside the fused GEMM epilogues)"""
    lib = _lib.load()
    _chk2(inp), _chk2(dout), _chk2(din)
    M, C2 = inp.shape
    _lib.check(lib.sdlt_geglu_bwd(_p(inp), _ld(inp), _p(dout), _ld(dout), M, C2 // 2, _p(din), _ld(din), _stream()), "sdlt_geglu_bwd")
    return din


MAP_SILU, MAP_DSILU, MAP_ADD, MAP_GELU, MAP_DGELU, MAP_QGELU, MAP_DQGELU = range(7)


def map_bf16(op, x, dy, y):
    lib = _lib.load()
    _chk2(x), _chk2(y)
    assert x.is_contiguous() and y.is_contiguous() and (dy is None or dy.is_contiguous())
    _lib.check(lib.sdlt_map_bf16(op, _p(x), _p(dy), _p(y), x.numel(), _stream()), "sdlt_map_bf16")
    return y


def timestep_embedding(t, out):
    """t fp32 [rows], out bf16 [rows, dim] = [cos | sin]."""
    lib = _lib.load()
    _chk2(t, F32), _chk2(out)
    _lib.check(lib.sdlt_timestep_embedding(_p(t), out.shape[0], out.shape[1], _p(out), _ld(out), _stream()), "sdlt_timestep_embedding")
    return out


def add_noise_nhwc(x0, noise, timesteps, alphas_cumprod, out, noisy_nchw=None):
    lib = _lib.load()
    B, Cc, H, W = x0.shape
    _chk2(x0, F32), _chk2(noise, F32), _chk2(alphas_cumprod, F32), _chk2(out)
    assert timesteps.dtype == torch.int64 and x0.is_contiguous() and noise.is_contiguous() and out.is_contiguous()
    _lib.check(lib.sdlt_add_noise_nhwc(_p(x0), _p(noise), _p(timesteps), _p(alphas_cumprod), B, Cc, H * W, out.shape[1], _p(out),
                                       _p(noisy_nchw), _stream()), "sdlt_add_noise_nhwc")
    return out


def masked_mse_fwd_bwd(pred, noise, noisy, mask, timesteps, alphas_cumprod, sums, loss_out, dpred, *, snr_gamma, v_prediction=False,
                       loss_scale=1.0):
    lib = _lib.load()
    B, Cc, H, W = noise.shape
    _chk2(pred, F32), _chk2(noise, F32), _chk2(mask, F32), _chk2(sums, F32), _chk2(loss_out, F32), _chk2(dpred)
    assert mask.is_contiguous() and noise.is_contiguous() and dpred.is_contiguous()
    _lib.check(lib.sdlt_masked_mse_fwd_bwd(_p(pred), _ld(pred), _p(noise), _p(noisy), _p(mask), _p(timesteps), _p(alphas_cumprod),
                                           B, Cc, H * W, dpred.shape[1], float(snr_gamma or 0.0), int(v_prediction), float(loss_scale),
                                           _p(sums), sums.numel(), _p(loss_out), _p(dpred), _stream()), "sdlt_masked_mse_fwd_bwd")


def adamw_fused(p, g, m, v, hyper, l1_sum=None):
    lib = _lib.load()
    for t in (p, g, m, v, hyper):
        _chk2(t, F32)
        assert t.is_contiguous()
    _lib.check(lib.sdlt_adamw_fused(_p(p), _p(g), _p(m), _p(v), p.numel(), _p(hyper), _p(l1_sum), _stream()), "sdlt_adamw_fused")


PRODIGY_HYPER = ("lr", "beta1", "beta2", "beta3", "eps", "weight_decay", "d_coef", "growth_rate", "l1_coef", "grad_scale",
                 "use_bias_correction", "safeguard_warmup", "decouple")
PRODIGY_STATE = ("d", "d0", "d_max", "d_numerator", "d_denom", "d_hat", "k", "dlr", "active")


def prodigy_step(p, g, p0, m, v, s, hyper, state, acc, l1_sum=None):
    """One prodigyopt-1.0 step on a flat fp32 arena; hyper / state are device fp32 rows laid out as PRODIGY_HYPER /
    PRODIGY_STATE, acc is a device fp64[2] scratch (include/sdlt_kernels.h: sdlt_prodigy_step)."""
    lib = _lib.load()
    for t in (p, g, p0, m, v, s, hyper, state):
        _chk2(t, F32)
        assert t.is_contiguous()
    assert acc.dtype == torch.float64 and acc.numel() >= 2 and hyper.numel() >= len(PRODIGY_HYPER) and state.numel() >= len(PRODIGY_STATE)
    _lib.check(lib.sdlt_prodigy_step(_p(p), _p(g), _p(p0), _p(m), _p(v), _p(s), p.numel(), _p(hyper), _p(state), _p(acc),
                                     _p(l1_sum), _stream()), "sdlt_prodigy_step")


class ShadowPlan:
    """Descriptor table for sdlt_lora_shadow_refresh: fp32 arena tensors -> bf16 compute copies."""

    def __init__(self, entries, device):
        # entries: list of (offset, rows, cols, src_ld, dst or None, dstT or None)
        descs = (_lib.ShadowDesc * len(entries))()
        block_desc, block_first = [], []
        nb = 0
        self.keep = []
        for i, (off, rows, cols, src_ld, dst, dstT) in enumerate(entries):
            d = descs[i]
            d.offset, d.rows, d.cols, d.src_ld = off, rows, cols, src_ld
            if dst is not None:
                _chk2(dst)
                d.dst, d.ld = dst.data_ptr(), _ld(dst)
            if dstT is not None:
                _chk2(dstT)
                d.dstT, d.ldT = dstT.data_ptr(), _ld(dstT)
            block_first.append(nb)
            blocks = ((rows + 63) // 64) * ((cols + 63) // 64)         # one workgroup per 64x64 tile
            block_desc += [i] * blocks
            nb += blocks
            self.keep += [dst, dstT]
        self.n_blocks = nb
        self.descs_dev = torch.frombuffer(bytearray(bytes(descs)), dtype=torch.uint8).to(device)
        self.block_desc_dev = torch.tensor(block_desc, dtype=torch.int32, device=device)
        self.block_first_dev = torch.tensor(block_first, dtype=torch.int32, device=device)

    def run(self, arena):
        lib = _lib.load()
        _chk2(arena, F32)
        _lib.check(lib.sdlt_lora_shadow_refresh(_p(self.descs_dev), _p(self.block_desc_dev), _p(self.block_first_dev), self.n_blocks,
                                                _p(arena), _stream()), "sdlt_lora_shadow_refresh")


class LnFoldPlan:
    """Descriptor table of sdlt_ln_fold_adapters: for every rank-16 adapter behind a folded LayerNorm, Ag = bf16(A o gamma) and the constants
    cA | abeta the consumer GEMM needs; one launch after every optimizer step (LoraArena.refresh_shadows)."""

    def __init__(self, items, device):
        # items: dict(A32 fp32 [rank, K] view, gamma, beta fp32 [K], Ag bf16 [16, K] view, consts fp32 [32] view)
        descs = (_lib.LnFoldDesc * len(items))()
        self.keep = items
        for d, it in zip(descs, items):
            A, Ag = it["A32"], it["Ag"]
            assert A.dtype == F32 and A.stride(1) == 1 and Ag.dtype == BF16 and Ag.stride(1) == 1 and Ag.shape[0] == 16 and A.shape[1] % 4 == 0
            d.A32, d.lda, d.gamma, d.beta, d.Ag, d.ldag = A.data_ptr(), A.stride(0), it["gamma"].data_ptr(), it["beta"].data_ptr(), Ag.data_ptr(), Ag.stride(0)
            d.consts, d.rank, d.K = it["consts"].data_ptr(), A.shape[0], A.shape[1]
        self.n = len(items)
        self.dev = _to_dev(descs, device)

    def run(self):
        _lib.check(_lib.load().sdlt_ln_fold_adapters(_p(self.dev), self.n, _stream()), "sdlt_ln_fold_adapters")


def _block_table(counts, device):
    block_desc, block_first, nb = [], [], 0
    for i, c in enumerate(counts):
        block_first.append(nb)
        block_desc += [i] * c
        nb += c
    return nb, torch.tensor(block_desc, dtype=torch.int32, device=device), torch.tensor(block_first, dtype=torch.int32, device=device)


def _to_dev(arr, device):
    return torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(device)


class DoraPlan:
    """Descriptor tables of the three batched DoRA launches (sdlt_dora_refresh / _scale_wt / _mag_grad) over every adapted layer.
    layers: dict(W [N,K] bf16, A_s [Rp,K], B_s [N,Rp], B32 fp32 [N,rank], mag fp32 [N], scale fp32 [N], Bt bf16 [Rp,N] view or None, s)
    wts   : dict(src, dst bf16 [rows, cols] (same strides), scale fp32 [period'], period, nvalid)
    grads : dict(dY, Y bf16 [M,N], bias fp32 [N] or None, mag, scale, gmag fp32 [N], gB fp32 [N,rank][, accumulate: gmag += and gB untouched - the
            second pass of step.TrainStep._tok_cond_reg, whose plan holds no layers / wts and runs behind the first pass's mag_grad])"""

    def __init__(self, layers, wts, grads, rank, Rp, device):
        self.rank, self.Rp, self.device = rank, Rp, device
        self.keep = [layers, wts, grads]
        arr = (_lib.DoraDesc * max(len(layers), 1))()
        for d, L in zip(arr, layers):
            W, A, B = L["W"], L["A_s"], L["B_s"]
            _chk2(W), _chk2(A), _chk2(B), _chk2(L["B32"], F32), _chk2(L["mag"], F32), _chk2(L["scale"], F32)
            N, K = W.shape
            assert K % 32 == 0 and tuple(A.shape) == (Rp, K) and tuple(B.shape) == (N, Rp) and tuple(L["B32"].shape) == (N, rank), (W.shape, A.shape, B.shape)
            d.W, d.ldw, d.A, d.lda, d.B, d.ldb = _p(W), _ld(W), _p(A), _ld(A), _p(B), _ld(B)
            d.mag, d.scale, d.B32, d.ldb32 = _p(L["mag"]), _p(L["scale"]), _p(L["B32"]), _ld(L["B32"])
            if L.get("Bt") is not None:
                _chk2(L["Bt"])
                assert tuple(L["Bt"].shape) == (Rp, N)
                d.Bt, d.ldbt = _p(L["Bt"]), _ld(L["Bt"])
            d.N, d.K, d.Rp, d.rank, d.s = N, K, Rp, rank, float(L["s"])
        self.layers_dev = _to_dev(arr, device)
        self.nb_layers, self.bd_layers, self.bf_layers = _block_table([(L["W"].shape[0] + 63) // 64 for L in layers], device)
        self.set_wts(wts)
        self.set_grads(grads)

    def set_wts(self, wts):
        self.keep[1] = wts
        arr = (_lib.DoraWtDesc * max(len(wts), 1))()
        for d, w in zip(arr, wts):
            src, dst = w["src"], w["dst"]
            _chk2(src), _chk2(dst), _chk2(w["scale"], F32)
            assert src.shape == dst.shape and src.stride() == dst.stride()
            d.src, d.dst, d.ld, d.scale = _p(src), _p(dst), _ld(src), _p(w["scale"])
            d.rows, d.cols, d.period, d.nvalid = src.shape[0], src.shape[1], w["period"], w["nvalid"]
        self.wts_dev = _to_dev(arr, self.device)
        self.nb_wts, self.bd_wts, self.bf_wts = _block_table([((w["src"].shape[0] + 31) // 32) * ((w["src"].shape[1] + 255) // 256) for w in wts], self.device)

    def set_grads(self, grads):
        self.keep[2] = grads
        arr = (_lib.DoraGradDesc * max(len(grads), 1))()
        off, c1, c2 = 0, [], []
        for d, g in zip(arr, grads):
            dY, Y = g["dY"], g["Y"]
            _chk2(dY), _chk2(Y), _chk2(g["gmag"], F32), _chk2(g["gB"], F32)
            M, N = Y.shape
            assert tuple(dY.shape[:1]) == (M,) and dY.shape[1] >= N and N % 8 == 0 and g["gB"].is_contiguous()
            assert dY.data_ptr() % 16 == 0 and Y.data_ptr() % 16 == 0 and _ld(dY) % 8 == 0 and _ld(Y) % 8 == 0
            d.dY, d.lddy, d.Y, d.ldy = _p(dY), _ld(dY), _p(Y), _ld(Y)
            if g.get("bias") is not None:
                _chk2(g["bias"], F32)
                d.bias = _p(g["bias"])
            d.mag, d.scale, d.gmag, d.gB = _p(g["mag"]), _p(g["scale"]), _p(g["gmag"]), _p(g["gB"])
            d.M, d.N, d.rank, d.splits, d.grad_scale = M, N, self.rank, max(1, min(64, (M + 2047) // 2048)), 1.0
            d.accumulate = int(bool(g.get("accumulate")))
            d.ws_off = off
            off += d.splits * 2 * N
            c1.append(((N + 63) // 64) * d.splits)
            c2.append((N + 255) // 256)
        self.grads_dev = _to_dev(arr, self.device)
        self.nb_g1, self.bd_g1, self.bf_g1 = _block_table(c1, self.device)
        self.nb_g2, self.bd_g2, self.bf_g2 = _block_table(c2, self.device)
        self.ws = torch.empty(max(off, 1), dtype=F32, device=self.device)

    def refresh(self, init=False):
        """scale = mag / ||W + s B A||_row (init: mag := the norm first), scaled B^T operands, then the scaled dX weights."""
        lib = _lib.load()
        if self.nb_layers:
            _lib.check(lib.sdlt_dora_refresh(_p(self.layers_dev), _p(self.bd_layers), _p(self.bf_layers), self.nb_layers, self.Rp, int(init), _stream()),
                       "sdlt_dora_refresh")
        self.scale_wts()

    def scale_wts(self):
        if self.nb_wts:
            _lib.check(_lib.load().sdlt_dora_scale_wt(_p(self.wts_dev), _p(self.bd_wts), _p(self.bf_wts), self.nb_wts, _stream()), "sdlt_dora_scale_wt")

    def mag_grad(self):
        if self.nb_g1:
            _lib.check(_lib.load().sdlt_dora_mag_grad(_p(self.grads_dev), _p(self.bd_g1), _p(self.bf_g1), self.nb_g1, _p(self.bd_g2), _p(self.bf_g2),
                                                      self.nb_g2, _p(self.ws), _stream()), "sdlt_dora_mag_grad")


def _shadow_adamw(self, p, g, m, v, hyper):
    """AdamW step fused into the refresh tiles (sdlt_adamw_shadow_refresh); covers exactly the elements the plan's descriptors tile."""
    lib = _lib.load()
    for t in (p, g, m, v, hyper):
        _chk2(t, F32)
    _lib.check(lib.sdlt_adamw_shadow_refresh(_p(self.descs_dev), _p(self.block_desc_dev), _p(self.block_first_dev), self.n_blocks,
                                             _p(p), _p(g), _p(m), _p(v), _p(hyper), _stream()), "sdlt_adamw_shadow_refresh")


ShadowPlan.adamw = _shadow_adamw


def dynamic_code_book(signed):
    """The 256-entry "dynamic" code book of bitsandbytes' 8-bit optimizers (bitsandbytes 0.43.1 functional.create_dynamic_map with its defaults: 7 exponent
    bits, 8 bits in all), sorted: decade i = 0..6 holds 2^i (signed; 2^(i+1) unsigned) values - the midpoints of an even split of [0.1, 1] - scaled by
    10^(i-6), mirrored for the signed book, plus 0 and 1.  The fp32 values follow the recipe's own arithmetic (linspace, means and the scaling all in fp32) so that the indices mean the same numbers as in the reference's optimizer state."""
    vals = []
    for i in range(7):
        edges = torch.linspace(0.1, 1, (2 ** i if signed else 2 ** (i + 1)) + 1)
        scaled = (10 ** (i - 6)) * ((edges[:-1] + edges[1:]) / 2.0)
        vals += scaled.tolist()
        if signed:
            vals += (-scaled).tolist()
    vals += [0.0, 1.0]
    assert len(vals) == 256
    return torch.tensor(sorted(vals), dtype=F32)


def q8_tables(device):
    """q1 | mid1 | q2 | mid2 for sdlt_adamw8_shadow_refresh (mid[k] = (q[k] + q[k+1]) / 2 in fp32, mid[255] = +inf)."""
    parts = []
    for signed in (True, False):
        q = dynamic_code_book(signed)
        parts += [q, torch.cat([(q[:-1] + q[1:]) * 0.5, torch.tensor([float("inf")])])]
    return torch.cat(parts).to(device)


def _shadow_adamw8(self, p, g, m8, v8, absmax, tables, hyper):
    """AdamW8bit step fused into the refresh tiles (sdlt_adamw8_shadow_refresh): one block of 2048 moments = half a tile; m8 / v8 tile-major, uint8 [n_blocks * 4096]."""
    lib = _lib.load()
    for t in (p, g, absmax, tables, hyper):
        _chk2(t, F32)
    assert m8.dtype == torch.uint8 and v8.dtype == torch.uint8 and m8.numel() >= 4096 * self.n_blocks and v8.numel() >= 4096 * self.n_blocks      # tile-major codes
    assert absmax.numel() == 4 * self.n_blocks and tables.numel() == 1024 and absmax.is_contiguous() and tables.is_contiguous()
    _lib.check(lib.sdlt_adamw8_shadow_refresh(_p(self.descs_dev), _p(self.block_desc_dev), _p(self.block_first_dev), self.n_blocks, _p(p), _p(g), _p(m8), _p(v8),
                                              _p(absmax), _p(tables), _p(hyper), _stream()), "sdlt_adamw8_shadow_refresh")


ShadowPlan.adamw8 = _shadow_adamw8


def adamw8_flat(p, g, m8, v8, absmax, tables, hyper):
    """AdamW8bit on a flat fp32 range (sdlt_adamw8_flat): blocks of 2048 consecutive elements, absmax fp32 [ceil(n / 2048), 2]."""
    lib = _lib.load()
    for t in (p, g, absmax, tables, hyper):
        _chk2(t, F32)
        assert t.is_contiguous()
    n = p.numel()
    assert g.numel() == n and m8.dtype == torch.uint8 and v8.dtype == torch.uint8 and m8.numel() == n and v8.numel() == n and m8.is_contiguous() and v8.is_contiguous()
    assert absmax.numel() == 2 * ((n + 2047) // 2048) and tables.numel() == 1024
    _lib.check(lib.sdlt_adamw8_flat(_p(p), _p(g), _p(m8), _p(v8), _p(absmax), n, _p(tables), _p(hyper), _stream()), "sdlt_adamw8_flat")


def add2d(a, b, out):
    lib = _lib.load()
    _chk2(a), _chk2(b), _chk2(out)
    M, Cc = a.shape
    _lib.check(lib.sdlt_add2d(_p(a), _ld(a), _p(b), _ld(b), _p(out), _ld(out), M, Cc, _stream()), "sdlt_add2d")
    return out


def sum2x2(inp, out, *, B, H, W):
    lib = _lib.load()
    _chk2(inp), _chk2(out)
    assert inp.is_contiguous() and out.is_contiguous()
    _lib.check(lib.sdlt_sum2x2(_p(inp), B, H, W, inp.shape[1], _p(out), _stream()), "sdlt_sum2x2")
    return out


class ColsumFinishPlan:
    """Descriptor table of sdlt_colsum_finish_batch: items = [(ws fp32 [nsplit * n], nsplit, out fp32 or bf16 [n])] - all reductions in one launch."""

    def __init__(self, items, device):
        descs = (_lib.ColsumFinishDesc * len(items))()
        self.keep, self.max_n = items, 0
        for d, (ws, nsplit, out) in zip(descs, items):
            _chk2(ws, F32)
            assert ws.is_contiguous() and out.is_contiguous() and out.is_cuda and out.dtype in (F32, BF16) and ws.numel() >= nsplit * out.numel()
            d.ws, d.nsplit, d.n = ws.data_ptr(), nsplit, out.numel()
            if out.dtype == F32:
                d.out32 = out.data_ptr()
            else:
                d.out16 = out.data_ptr()
            self.max_n = max(self.max_n, out.numel())
        self.n = len(items)
        self.descs_dev = torch.frombuffer(bytearray(bytes(descs)), dtype=torch.uint8).to(device)

    def run(self):
        _lib.check(_lib.load().sdlt_colsum_finish_batch(_p(self.descs_dev), self.n, self.max_n, _stream()), "sdlt_colsum_finish_batch")


_colsum_scratch = {}


def colsum(x, out, *, B, R):
    """out[b, c] = sum over the R rows of batch b of x[:, c]; out is fp32 or bf16 [B, C]."""
    lib = _lib.load()
    _chk2(x)
    assert out.is_cuda and out.dtype in (F32, BF16) and out.is_contiguous()
    ws, _ = gn_workspace(x.device)          # the same per-(device, owner) fp32 scratch as the GroupNorm reductions: only live inside a call
    f, h = (_p(out), C.c_void_p(0)) if out.dtype == F32 else (C.c_void_p(0), _p(out))
    _lib.check(lib.sdlt_colsum(_p(x), _ld(x), B, R, x.shape[1], _p(ws), ws.numel(), f, h, _stream()), "sdlt_colsum")
    return out


def embed_gather(table, ids, pos, out, *, B, T, Tp):
    """out[b*Tp + t] = table[ids[b, t]] + pos[t]; rows T..Tp-1 of every batch are zero."""
    lib = _lib.load()
    _chk2(table), _chk2(pos), _chk2(out)
    assert ids.dtype == torch.int64 and ids.is_contiguous() and ids.numel() == B * T and out.shape[0] == B * Tp
    _lib.check(lib.sdlt_embed_gather(_p(table), _ld(table), _p(ids), _p(pos), _ld(pos), B, T, Tp, table.shape[1], _p(out), _ld(out), _stream()),
               "sdlt_embed_gather")
    return out


def embed_grad(dx, ids, train_ids, grad, *, B, T, Tp, accumulate=False):
    """grad[j] (+)= sum of dx rows whose token id equals train_ids[j] (fp32 [n, D])."""
    lib = _lib.load()
    _chk2(dx), _chk2(grad, F32)
    assert ids.dtype == torch.int64 and train_ids.dtype == torch.int64 and grad.is_contiguous()
    _lib.check(lib.sdlt_embed_grad(_p(dx), _ld(dx), _p(ids), _p(train_ids), train_ids.numel(), B, T, Tp, dx.shape[1], _p(grad), int(accumulate),
                                   _stream()), "sdlt_embed_grad")
    return grad


def ti_std_reg(rows, grad, loss_out, *, target_mean, target_var, weight):
    """loss_out += weight * mean_j (target_mean - std(rows_j))^2 / target_var ; grad += its gradient (fp32 [n, D])."""
    lib = _lib.load()
    _chk2(rows, F32), _chk2(grad, F32), _chk2(loss_out, F32)
    assert rows.is_contiguous() and grad.is_contiguous()
    _lib.check(lib.sdlt_ti_std_reg(_p(rows), rows.shape[0], rows.shape[1], float(target_mean), float(target_var), float(weight), _p(grad),
                                   _p(loss_out), _stream()), "sdlt_ti_std_reg")
