"""UNet2DCondition (SD1.5 / SDXL) with rank-r LoRA adapters as an explicit forward/backward plan over
the HIP kernels of libsdlt_kernels.so.

This is the drop-in for the model call of the reference's training step
(/root/reference main.py:329-336: `unet(noisy_latent, timesteps, encoder_hidden_states=..., added_cond_kwargs=...)`
with the peft adapters of trainer/optimizer.py:84-95 and the DAAM hooks of trainer/ti_cross_attn_loss.py:336-364)
and for `loss.backward()` (main.py:363) restricted to what the reference actually trains: LoRA A/B and the
gradient w.r.t. the text conditioning (which carries the textual-inversion gradient).

MI355X-first choices:
  * no autograd tape, no tracing compiler: the topology is static, so every layer owns persistent buffers
    (activations are kept, never recomputed - 288 GB HBM) and an explicit backward; the whole step is
    captured once into a hipGraph and replayed.
  * NHWC / token-major bf16 activations: a feature map IS the [B*H*W, C] matrix the transformer blocks use,
    3x3 convs are implicit GEMMs, the up-block skip concat is never materialised.
  * frozen weights are stored in both orientations (W for forward, W^T for dX).
  * every LoRA weight gradient of the step is produced by ONE grouped launch at the end of backward.

Weight names follow the diffusers state dict so real checkpoints and the kohya exporter line up.
"""
import copy
import os
import math

import torch

from . import ops as _ops
from .topology import CONFIGS, TIME_DIM_MULT, lora_targets, param_shapes  # noqa: F401

BF16, F32 = torch.bfloat16, torch.float32


def _pad_to(x, m):
    return (x + m - 1) // m * m


XATTN_WGS = int(os.environ.get("SDLT_XATTN_WGS", "160"))     # workgroups of a cross-attention backward launch (query splits x heads x batch)

class Runtime:
    """Execution context shared by all layers: device, batch, activation dtype and the op table."""

    def __init__(self, device, B, act_dtype=BF16, ops=None):
        self.device, self.B, self.act = torch.device(device), B, act_dtype
        self.ops = ops if ops is not None else _ops
        self.lora_problems = []      # filled by layers on their first backward
        self.daam = []               # (name, scores fp32 [B, N, CTX_PAD]) of hooked attn2 layers, reference order (keep_daam_maps)
        self.keep_daam_maps = False  # per-layer maps are only needed for the reference's debug heat-map plots
        self.daam_sums = {}          # tokens-per-image N -> (fp32 [B*N, CTX_PAD] sum of the hooked layers' raw scores, n_layers)
        self.daam_grads = None       # N -> (dS bf16 [B*N, CTX_PAD], dS^T bf16 [B*CTX_PAD, N]) set by the token-attention loss
        self.want_dpooled = False    # SDXL: back-propagate into the pooled text embedding (textual inversion)
        self.dsemb = None
        self.daam_layer_grads = None  # attention name -> (dS, dS^T) of ONE hooked layer (shim: autograd hands every layer its own score gradient)
        self.daam_applied = False    # this step's score-gradient GEMMs were issued up front (UNet.daam_backward)
        self.defer_daam_scores = True  # the hooked layers' score GEMMs run batched after the forward pass (UNet._daam_forward)
        self.trainer = None          # fullft.WeightTrainer when the whole UNet is trained (is_lora = False)
        self._scratch = {}
        # GroupNorm statistics (forward sums, backward sums) of all layers live in two small arenas: the UNet clears each with one
        # fill per pass (gn_prezero) instead of one zero launch per layer (92 per SDXL step)
        self._gn_arena = {"fwd": [None, 0], "bwd": [None, 0]}
        self.gn_prezero = False

    def gn_stats(self, which, n):
        ar = self._gn_arena[which]
        if ar[0] is None:
            ar[0] = torch.zeros(1 << 18, dtype=F32, device=self.device)      # 256 K floats: 2048 GroupNorm calls at batch 2
        assert ar[1] + n <= ar[0].numel(), "GroupNorm statistics arena exhausted"
        v = ar[0][ar[1]: ar[1] + n]
        ar[1] += n
        return v

    def gn_clear(self, which):
        """(no-op: the GroupNorm statistics are produced by a fixed-order two-stage reduction that overwrites its output -
        sdlt_groupnorm_fwd / _bwd - so nothing has to be cleared between passes any more)"""

    def scratch(self, key, nfloats):
        """fp32 scratch shared by every layer: valid only inside the op call that receives it (all ops run on one stream)."""
        t = self._scratch.get(key)
        if t is None or t.numel() < nfloats:
            t = torch.empty(nfloats, dtype=F32, device=self.device)
            self._scratch[key] = t
        return t[:nfloats]

    def empty(self, *shape, dtype=None):
        return torch.empty(*shape, dtype=dtype or self.act, device=self.device)

    def zeros(self, *shape, dtype=None):
        return torch.zeros(*shape, dtype=dtype or self.act, device=self.device)


class _Module:
    def __init__(self, rt, name):
        self.rt, self.name, self._b = rt, name, {}

    def buf(self, key, *shape, dtype=None, zero=False):  # persistent, owned by this module
        t = self._b.get(key)
        if t is None:
            t = (self.rt.zeros if zero else self.rt.empty)(*shape, dtype=dtype)
            self._b[key] = t
        return t


class ArenaView:
    """An adapter arena as seen by a plan clone: same parameters, shadows and gradient tensors, but its OWN list of LoRA-gradient
    problems (the clone's dY / T / U buffers), to be run as a second, accumulating grouped launch."""

    def __init__(self, arena):
        self._arena, self.problems, self.dora_grads = arena, [], []

    def __getattr__(self, k):
        return getattr(self._arena, k)


def clone_plan(mod, rt, _memo=None, arena=None):
    """Structural copy of an execution plan for ANOTHER batch size: the clone shares every weight tensor with `mod` but owns
    its activation buffers and runs on `rt` (used for the text encoders' second, 4-caption pass of the tok_cond_reg_w term).
    arena: an ArenaView that replaces the adapter arena in the clone's layers (text-encoder LoRA)."""
    memo = {} if _memo is None else _memo
    if id(mod) in memo:
        return memo[id(mod)]
    c = copy.copy(mod)
    memo[id(mod)] = c
    c.rt, c._b = rt, {}
    if arena is not None and getattr(c, "arena", None) is not None:
        c.arena = arena
    if hasattr(c, "_registered"):
        c._registered = False
    for k, v in list(vars(c).items()):
        if isinstance(v, _Module):
            setattr(c, k, clone_plan(v, rt, memo, arena))
        elif isinstance(v, (list, tuple)) and v and all(isinstance(x, _Module) for x in v):
            setattr(c, k, type(v)(clone_plan(x, rt, memo, arena) for x in v))
    return c


# ---------------------------------------------------------------------------------------- LoRA arena

class LoraArena:
    """Flat fp32 master copy of every LoRA A/B (plus grads and AdamW moments) and the bf16 compute copies."""

    def __init__(self, rt, rank, alpha_multiplier=1.0, problems=None, dora=False):
        """dora: weight-decomposed adapters (peft use_dora, trainer/optimizer.py:86-95): every entry also owns a trained magnitude
        vector M [N]; y = (M / ||W + s B A||_row) * (x W^T + s x A^T B^T) + bias with the norm detached.  The per-column factor
        (`scale`), the scaled backward operands and M's gradient are maintained by three batched launches (ops.DoraPlan)."""
        self.rt, self.rank, self.dora = rt, rank, bool(dora)
        self.dora_wts, self.dora_grads, self.dora_plan, self._dora_ngrads = [], [], None, 0
        # LoRA-gradient problems of the adapters in this arena, appended by the layers on their first backward.  The UNet
        # arena shares the runtime's list; the text-encoder arena (step.TextStack) keeps its own, run after the text backward.
        self.problems = rt.lora_problems if problems is None else problems
        self.Rp = 16 if rank <= 16 else (32 if rank <= 32 else 64)
        assert rank <= 64, "LoRA rank > 64 not supported by the fused kernels"
        self.scale = (rank * alpha_multiplier) / rank  # peft: lora_alpha / r, lora_alpha = r * multiplier (optimizer.py:88)
        self.entries = []   # dict(name, kind, offA, offB, N, K, conv)
        self.n = 0
        self._shadow_entries = []
        self.params = self.grads = self.m = self.v = None
        self.ln_items, self._ln_plan = [], None      # adapters behind a folded LayerNorm (Linear.fold_ln): Ag = A o gamma + constants, refreshed with the shadows

    def add(self, name, N, K, conv_cin=None, W=None):
        """W: the layer's bf16 forward operand [N, K] (DoRA: its rows enter the norm)."""
        r, Rp, rt = self.rank, self.Rp, self.rt
        e = dict(name=name, N=N, K=K, conv_cin=conv_cin, offA=self.n, offB=self.n + r * K, W=W)
        self.n += r * K + N * r
        if self.dora:
            assert W is not None and N % 4 == 0
            e["offM"] = self.n
            self.n += N
        e["A_s"] = rt.zeros(Rp, K)          # LoRA-down, forward orientation   [Rp, K]
        e["B_s"] = rt.zeros(N, Rp)          # LoRA-up                          [N, Rp]
        e["Bt_s"] = rt.zeros(Rp, N)         # backward LoRA-down operand       [Rp, N]
        if conv_cin is None:
            e["At_s"] = rt.zeros(K, Rp)     # backward LoRA-up operand         [K, Rp]
        else:
            e["Ab_s"] = rt.zeros(conv_cin, 9 * 64)  # dX weights of the 3x3 LoRA-down conv: [Cin, tap*64 + rank]
        self.entries.append(e)
        return e

    def finalize(self):
        rt = self.rt
        self.params = rt.zeros(self.n, dtype=F32)
        self.grads = rt.zeros(self.n, dtype=F32)
        self.m = rt.zeros(self.n, dtype=F32)
        self.v = rt.zeros(self.n, dtype=F32)
        sh = []
        r = self.rank
        if self.dora:
            self.dscale = rt.zeros(sum(e["N"] for e in self.entries), dtype=F32)
            off = 0
            for e in self.entries:
                e["offS"] = off
                e["scale"] = self.dscale[off: off + e["N"]]
                off += e["N"]
        for e in self.entries:
            N, K = e["N"], e["K"]
            if e["conv_cin"] is None:
                sh.append((e["offA"], r, K, K, e["A_s"], e["At_s"]))   # (offset, rows, cols, src_ld, dst, dstT)
            else:
                cin = e["conv_cin"]
                sh.append((e["offA"], r, K, K, e["A_s"], None))
                for tap in range(9):   # A[:, tap, :] ([r, Cin], row stride 9*Cin) -> Ab[ci, tap*64 + rank]
                    sh.append((e["offA"] + tap * cin, r, cin, K, None, e["Ab_s"][:, tap * 64: tap * 64 + 64]))
            sh.append((e["offB"], N, r, r, e["B_s"], None if self.dora else e["Bt_s"]))      # DoRA: B^T is written scaled by the refresh
            e["A"] = self.params[e["offA"]: e["offA"] + r * K].view(r, K)
            e["B"] = self.params[e["offB"]: e["offB"] + N * r].view(N, r)
            e["gA"] = self.grads[e["offA"]: e["offA"] + r * K].view(r, K)
            e["gB"] = self.grads[e["offB"]: e["offB"] + N * r].view(N, r)
            if self.dora:
                e["M"] = self.params[e["offM"]: e["offM"] + N]
                e["gM"] = self.grads[e["offM"]: e["offM"] + N]
        self._shadow_plan = rt.ops.ShadowPlan(sh, rt.device)
        if self.ln_items:
            for it in self.ln_items:
                it["A32"] = it["entry"]["A"]
            self._ln_plan = rt.ops.LnFoldPlan(self.ln_items, rt.device)
            self._ln_plan.run()
        if self.dora:
            self._build_dora_plan()
            self.dora_plan.refresh(init=True)        # peft dora_init: magnitude = ||W + s B A||_row of the freshly injected adapter (B = 0)

    def _build_dora_plan(self):
        layers = [dict(W=e["W"], A_s=e["A_s"], B_s=e["B_s"], B32=e["B"], mag=e["M"], scale=e["scale"], Bt=e["Bt_s"], s=self.scale) for e in self.entries]
        wts = []
        for w in self.dora_wts:
            ents = w["entries"]
            for a, b in zip(ents, ents[1:]):
                assert b["offS"] == a["offS"] + a["N"], "stacked projections must be consecutive arena entries"
            n = sum(e_["N"] for e_ in ents)
            wts.append(dict(src=w["src"], dst=w["dst"], scale=self.dscale[ents[0]["offS"]: ents[0]["offS"] + n], period=w.get("period", n), nvalid=w.get("nvalid", n)))
        self.dora_plan = self.rt.ops.DoraPlan(layers, wts, self.dora_grads, self.rank, self.Rp, self.rt.device)
        self._dora_ngrads = len(self.dora_grads)

    def scale_of(self, entries):
        """fp32 view of the DoRA column factors of consecutive entries (one entry: a layer; several: a stack of projections)."""
        e0 = entries[0]
        return self.dscale[e0["offS"]: e0["offS"] + sum(e["N"] for e in entries)]

    def set_scale(self, scale):
        """lora_alpha / r in effect (the validation renders run the adapters at a reduced scale, checkpoint.py:31-55)."""
        if scale == self.scale:          # (the sampler sets the render scale on every call: nothing to rebuild when it has not changed)
            return
        self.scale = scale
        if self.dora:
            self._build_dora_plan()
            self.dora_plan.refresh()

    def refresh_shadows(self):
        self._shadow_plan.run(self.params)
        if self._ln_plan is not None:
            self._ln_plan.run()
        if self.dora:
            self.dora_plan.refresh()

    def dora_mag_grad(self):
        """Gradient of the magnitudes and the row scaling of dB; after the grouped LoRA-gradient launch of a backward pass."""
        if not self.dora:
            return
        if self._dora_ngrads != len(self.dora_grads):
            self.dora_plan.set_grads(self.dora_grads)
            self._dora_ngrads = len(self.dora_grads)
        self.dora_plan.mag_grad()

    # ---- host-side (load / export) in the reference's layouts -------------------------------
    def load(self, lora_dict):
        """lora_dict: module -> (A, B) in peft layout (conv: A [r,Cin,3,3], B [Cout,r,1,1])."""
        for e in self.entries:
            A, B, *rest = lora_dict[e["name"]]
            if rest and self.dora:          # (A, B, magnitude); without it the magnitudes keep their value
                e["M"].copy_(rest[0].reshape(-1).to(self.rt.device, F32))
            if A.dim() == 4:
                A = A.permute(0, 2, 3, 1).reshape(A.shape[0], -1)   # [r, (tap, ci)]
                B = B.reshape(B.shape[0], -1)
            e["A"].copy_(A.to(self.rt.device, F32))
            e["B"].copy_(B.to(self.rt.device, F32))
        self.refresh_shadows()

    def export(self, which="params"):
        out = {}
        r_ = self.rank
        for e in self.entries:
            if which in ("m", "v"):          # AdamW moments, in the layout of the parameters they belong to
                flat = self.m if which == "m" else self.v
                A, B = flat[e["offA"]: e["offA"] + r_ * e["K"]].view(r_, e["K"]), flat[e["offB"]: e["offB"] + e["N"] * r_].view(e["N"], r_)
            else:
                A, B = (e["A"], e["B"]) if which == "params" else (e["gA"], e["gB"])
            A, B = A.detach().float().cpu(), B.detach().float().cpu()
            if e["conv_cin"] is not None:
                r = A.shape[0]
                A = A.reshape(r, 3, 3, e["conv_cin"]).permute(0, 3, 1, 2).contiguous()
                B = B.reshape(B.shape[0], r, 1, 1)
            out[e["name"]] = (A, B)
            if self.dora:                   # peft lora_magnitude_vector: [N] (Linear), [1, N, 1, 1] (Conv2d)
                m = ((self.m if which == "m" else self.v)[e["offM"]: e["offM"] + e["N"]] if which in ("m", "v") else (e["M"] if which == "params" else e["gM"])).detach().float().cpu().clone()
                out[e["name"]] = (A, B, m.reshape(1, -1, 1, 1) if e["conv_cin"] is not None else m)
        return out


# ---------------------------------------------------------------------------------------- leaf layers

def _mark_frozen(rt, *ws):
    """Frozen GEMM operands (no trainer; nothing rewrites them): the wave-split-K kernel may read them through a fragment-major copy (ops.wsk_mark_frozen)."""
    mark = getattr(rt.ops, "wsk_mark_frozen", None)
    if mark is not None and rt.trainer is None:
        for w in ws:
            if w is not None:
                mark(w)


class Linear(_Module):
    """y = x W^T + b (+ LoRA) (+ residual);  dx = dy W (+ LoRA) (+ dres)."""

    def __init__(self, rt, name, sd, arena=None, need_dx=True):
        super().__init__(rt, name)
        w = sd[name + ".weight"]
        if w.dim() == 4:   # 1x1 conv stored [Cout, Cin, 1, 1]
            w = w.reshape(w.shape[0], w.shape[1])
        self.N, self.K = w.shape
        self.W = w.to(rt.device, rt.act).contiguous()
        self.Wt = w.t().to(rt.device, rt.act).contiguous() if need_dx else None
        b = sd.get(name + ".bias")
        self.bias = b.to(rt.device, F32).contiguous() if b is not None else None
        self.lora = arena.add(name, self.N, self.K, W=self.W) if arena is not None else None
        self.arena = arena
        self.dora = arena is not None and arena.dora
        if self.dora and need_dx:          # dX operand with the DoRA column factor folded into its K axis (refreshed every step)
            self.Wt_d = torch.empty_like(self.Wt)
            self._dora_wt = dict(src=self.Wt, dst=self.Wt_d, entries=[self.lora])
            arena.dora_wts.append(self._dora_wt)
        # full fine-tune: the fp32 master of weight / bias lives in the trainer's arena (fullft.WeightTrainer)
        tr = rt.trainer if (rt.trainer is not None and rt.trainer.registering) else None
        self.trainer = tr
        if tr is not None:
            self.went = tr.add(name + ".weight", w.float(), "conv1x1" if sd[name + ".weight"].dim() == 4 else "matrix")
            self.bent = tr.add(name + ".bias", b.float(), "vector") if b is not None else None
            tr.on_finalize(self._bind_trainer)
        _mark_frozen(rt, self.W, self.Wt)      # (DoRA's dX operand is Wt_d, rewritten every step: never marked)

    def _bind_trainer(self):
        tr = self.trainer
        tr.shadow(self.went, self.N, self.K, self.K, self.W, self.Wt)     # bf16 W (and W^T) follow the master after each step
        if self.bent is not None:
            self.bias = tr.view(self.bent)

    ln = None

    def fold_ln(self, norm, w32, keep_plain=False):
        """The LayerNorm `norm` in front of this projection runs inside its GEMM from now on (sdlt_gemm_params.ln_c1 / sdlt_wsk_gemm_ln): the forward
        operand becomes W o gamma, the bias c2 = W beta + b; an adapter's LoRA-down rows become A o gamma (refreshed with the shadows).  The
        backward is untouched: it works on W^T / A^T and hands dL/dy to norm.backward as before.  w32: this layer's fp32 weight [N, K] in the
        row order of self.W.  keep_plain: the unfolded operand stays too (forward(..., unfolded=True) on the output of the LayerNorm launch) - for the
        consumer whose fold only pays when the producer left row partials (ff.net.0.proj, 26 MB per block of 288 GB)."""
        rt = self.rt
        assert self.trainer is None and not self.dora and (self.lora is None or self.arena.Rp == 16)
        if keep_plain:
            assert self.lora is None
            self.W_plain, self.bias_plain = self.W, self.bias
        self.W, self.ln_c1, self.bias = rt.ops.fold_layernorm(w32.to(rt.device), self.bias, norm.gamma, norm.beta, dtype=rt.act)
        _mark_frozen(rt, self.W)
        if self.lora is not None:
            self.lora["W"] = self.W
            self.ln_Ag, self.ln_consts = rt.zeros(self.arena.Rp, self.K), rt.zeros(32, dtype=F32)
            self.arena.ln_items.append(dict(entry=self.lora, gamma=norm.gamma, beta=norm.beta, Ag=self.ln_Ag, consts=self.ln_consts))
            norm.emit_y = True
        self.ln = norm
        norm.folded = True

    def weight_grad(self, dy, xs=None):
        self.trainer.linear(self.went, self.bent, xs if xs is not None else [self._x], dy)

    def forward(self, x, *, residual=None, Ct=None, key="y", out=None, train=True, geglu_out=None, act_out=None, parts_for=None, unfolded=False):
        """parts_for: the (folded) LayerNorm that reads this layer's output next - where the product runs on the wave-split-K kernel it also leaves
        that LayerNorm's row partials (ops.gemm ln_parts_out), so the consumer GEMM has no statistics to compute."""
        M = x.shape[0]
        y = out if out is not None else self.buf(key, M, self.N)
        lora, ln = None, None
        if self.lora is not None:
            T = self.buf("T", M, self.arena.Rp) if train else None
            lora = (self.lora["A_s"] if self.ln is None else self.ln_Ag, self.lora["B_s"], self.arena.scale, T)
            self._x = x if self.ln is None else self.ln.ybuf()     # (folded LayerNorm: the normalised rows are written by its backward)
        if self.trainer is not None:
            self._x = x
        W, bias = self.W, self.bias
        if unfolded and self.ln is not None:       # x is the OUTPUT of the LayerNorm launch (fold_ln keep_plain)
            W, bias = self.W_plain, self.bias_plain
        elif self.ln is not None:
            ln = (self.ln_c1, self.ln.stats_buf(), self.ln.eps, self.ln_consts if self.lora is not None else None) + (self.ln._parts or (None, 0))
        if self.dora:
            # y = scale * (x W^T + s x A^T B^T) + bias; the residual is added by a second launch because the magnitude gradient
            # needs the layer's own output (DoraPlan.mag_grad)
            assert geglu_out is None and act_out is None and not (residual is not None and Ct is not None)
            if residual is None:
                self.rt.ops.gemm(x, self.W, y, lora=lora, bias=self.bias, Ct=Ct, col_scale=self.lora["scale"])
                self._y0 = y
                return y
            # round 6: the product, the layer's own output y0 and y = y0 + residual in ONE launch where the wave-split-K kernel takes it (ops.gemm out0=; otherwise gemm + add2d inside
            # ops.gemm), and then also the row partials of the LayerNorm that reads y next
            y0 = self.buf(key + "0", M, self.N)
            kwd = {}
            if parts_for is not None:
                parts_for._parts = None
                if parts_for.folded and PARTS and hasattr(self.rt.ops, "gemm_emits_parts") and self.ln is None:
                    P = self.rt.ops.gemm_emits_parts(M, self.N, self.K, self.arena.Rp, self.W, dora=True)
                    if P:
                        parts_for._parts = (parts_for.buf("parts", M * P * 2, dtype=F32), P)
                        kwd["ln_parts_out"] = parts_for._parts[0]
            self.rt.ops.gemm(x, self.W, y, lora=lora, bias=self.bias, residual=residual, col_scale=self.lora["scale"], out0=y0, **kwd)
            self._y0 = y0
            return y
        kw = {} if ln is None else {"ln": ln}
        if parts_for is not None:
            parts_for._parts = None
            if (parts_for.folded and PARTS and hasattr(self.rt.ops, "gemm_emits_parts") and self.ln is None and Ct is None and geglu_out is None and act_out is None):
                P = self.rt.ops.gemm_emits_parts(M, self.N, self.K, self.arena.Rp if self.lora is not None else 0, W)
                if P:
                    parts_for._parts = (parts_for.buf("parts", M * P * 2, dtype=F32), P)
                    kw["ln_parts_out"] = parts_for._parts[0]
        if geglu_out is not None:
            self.rt.ops.gemm(x, W, y, lora=lora, bias=bias, residual=residual, Ct=Ct, geglu_out=geglu_out, **kw)
        elif act_out is not None:
            self.rt.ops.gemm(x, W, y, lora=lora, bias=bias, residual=residual, Ct=Ct, act_out=act_out, **kw)
        else:
            self.rt.ops.gemm(x, W, y, lora=lora, bias=bias, residual=residual, Ct=Ct, **kw)
        return y

    def backward(self, dy, *, dres=None, Ct=None, key="dx", out=None, dact_in=None, rowdot=None):
        """rowdot: see ops.gemm (attn1.to_out.0: the attention backward's row term as a side output of this dX product)."""
        M = dy.shape[0]
        dx = out if out is not None else self.buf(key, M, self.K)
        lora = None
        if self.lora is not None:
            U = self.buf("U", M, self.arena.Rp)
            lora = (self.lora["Bt_s"], self.lora["At_s"], self.arena.scale, U)
            if not getattr(self, "_registered", False):
                r = self.arena.rank
                self.arena.problems += [
                    dict(P=dy, Q=self._b["T"], out=self.lora["gB"], M=M, Cw=self.N, R=r, rank_major=False),
                    dict(P=self._x, Q=U, out=self.lora["gA"], M=M, Cw=self.K, R=r, rank_major=True)]
                if self.dora:
                    self.arena.dora_grads.append(dict(dY=dy, Y=self._y0, bias=self.bias, mag=self.lora["M"], scale=self.lora["scale"],
                                                      gmag=self.lora["gM"], gB=self.lora["gB"]))
                self._registered = True
        if self.trainer is not None:
            self.weight_grad(dy)
        Wt = self.Wt_d if self.dora else self.Wt
        if dact_in is not None:
            self.rt.ops.gemm(dy, Wt, dx, lora=lora, residual=dres, Ct=Ct, dact_in=dact_in)
        elif rowdot is not None:
            self.rt.ops.gemm(dy, Wt, dx, lora=lora, residual=dres, Ct=Ct, rowdot=rowdot)
        else:
            self.rt.ops.gemm(dy, Wt, dx, lora=lora, residual=dres, Ct=Ct)
        return dx


class StackedLinear(_Module):
    """Forward of several Linear layers that read the same input (to_q|to_k|to_v, to_k|to_v) as ONE GEMM over the
    stacked weights, each projection keeping its own LoRA adapter (sdlt_gemm_bf16 lora_group_n).  The members keep their
    own backward (dX needs per-projection dY anyway); this class only re-points their weights, LoRA shadows, outputs and
    T buffers at slices of the stacked tensors.  Must be built before LoraArena.finalize()."""

    def __init__(self, rt, name, members):
        super().__init__(rt, name)
        self.members = members
        N, K = members[0].N, members[0].K
        assert all(m.N == N and m.K == K for m in members), "stacked projections must share their shape"
        self.N, self.K, self.G = N, K, len(members)
        self.W = torch.cat([m.W for m in members], 0).contiguous()
        _mark_frozen(rt, self.W)
        for g, m in enumerate(members):
            m.W = self.W[g * N:(g + 1) * N]
            if m.lora is not None:
                m.lora["W"] = m.W
        self.bias = torch.cat([m.bias for m in members]).contiguous() if members[0].bias is not None else None
        self.arena = members[0].arena
        self.has_lora = members[0].lora is not None
        self.dora = self.has_lora and self.arena.dora
        self.trainer = members[0].trainer
        if self.trainer is not None:
            # the members' master weights are consecutive in the trainer's arena: one [G*N, K] gradient, one dW GEMM
            assert self.bias is None, "stacked projections with biases are not part of the UNet"
            e0 = members[0].went
            self.went = dict(name=name + ".weight", off=None, shape=(self.G * N, K), kind="matrix")

            def _bind_stack():          # arena offsets exist once the trainer is finalized
                for g, m in enumerate(members):
                    assert m.went["off"] == e0["off"] + g * N * K, "stacked members must be registered back to back"
                self.went["off"] = e0["off"]
            self.trainer.on_finalize(_bind_stack)
            self.Wt = self.W.t().contiguous()
            for g, m in enumerate(members):
                m.Wt = self.Wt[:, g * N:(g + 1) * N]
        if self.has_lora:
            Rp = self.arena.Rp
            self.A_cat, self.B_cat = rt.zeros(self.G * Rp, K), rt.zeros(self.G * N, Rp)
            for g, m in enumerate(members):   # the arena's shadow refresh now writes straight into the stacked operands
                m.lora["A_s"] = self.A_cat[g * Rp:(g + 1) * Rp]
                m.lora["B_s"] = self.B_cat[g * N:(g + 1) * N]
            # backward operands of the K-grouped dX GEMM: the members' B^T side by side, their A^T side by side
            self.kgrouped = self.G <= 4 and N % 64 == 0
            if self.kgrouped:
                self.Bt_cat, self.At_cat = rt.zeros(Rp, self.G * N), rt.zeros(K, self.G * Rp)
                for g, m in enumerate(members):
                    m.lora["Bt_s"] = self.Bt_cat[:, g * N:(g + 1) * N]
                    m.lora["At_s"] = self.At_cat[:, g * Rp:(g + 1) * Rp]
                if self.dora:       # one scaled dX operand for the stack instead of the members'
                    self.Wt = self.W.t().contiguous()
                    self.Wt_d = torch.empty_like(self.Wt)
                    for m in members:
                        if getattr(m, "_dora_wt", None) is not None:
                            self.arena.dora_wts[:] = [w for w in self.arena.dora_wts if w is not m._dora_wt]
                            m._dora_wt = m.Wt_d = None
                        m.Wt = None
                    self.arena.dora_wts.append(dict(src=self.Wt, dst=self.Wt_d, entries=[m.lora for m in members]))

    ln = None

    def fold_ln(self, norm, w32s):
        """Linear.fold_ln for the stack (w32s: the members' fp32 weights): one folded operand, one adapter-constant block per member."""
        rt = self.rt
        assert self.trainer is None and not self.dora and (not self.has_lora or self.arena.Rp == 16)
        N = self.N
        if getattr(self, "Wt", None) is None:      # the dX operand is built lazily from W: take it from the UNFOLDED weights now
            self.Wt = self.W.t().contiguous()
            _mark_frozen(self.rt, self.Wt)
            for m in self.members:
                m.Wt = None
        self.W, self.ln_c1, self.bias = rt.ops.fold_layernorm(torch.cat([w.to(rt.device) for w in w32s], 0), self.bias, norm.gamma, norm.beta, dtype=rt.act)
        _mark_frozen(rt, self.W)
        for g, m in enumerate(self.members):
            m.W = self.W[g * N:(g + 1) * N]
            if m.lora is not None:
                m.lora["W"] = m.W
        if self.has_lora:
            Rp = self.arena.Rp
            self.ln_Ag, self.ln_consts = rt.zeros(self.G * Rp, self.K), rt.zeros(self.G * 32, dtype=F32)
            for g, m in enumerate(self.members):
                self.arena.ln_items.append(dict(entry=m.lora, gamma=norm.gamma, beta=norm.beta, Ag=self.ln_Ag[g * Rp:(g + 1) * Rp], consts=self.ln_consts[g * 32:(g + 1) * 32]))
            norm.emit_y = True
        self.ln = norm
        norm.folded = True

    def prepare(self, x):
        """Allocates the stacked output / T buffers for input x and points the members at their slices (no launch)."""
        M, N, G = x.shape[0], self.N, self.G
        y = self.buf("y", M, G * N)
        T = None
        if self.has_lora:
            Rp = self.arena.Rp
            T = self.buf("T", M, G * Rp)
        outs = []
        xin = x if self.ln is None or not self.has_lora else self.ln.ybuf()      # (folded LayerNorm: the adapter gradients read the rows its backward writes)
        for g, m in enumerate(self.members):
            m._x = xin
            m._b["y"] = m._y0 = y[:, g * N:(g + 1) * N]
            if self.has_lora:
                m._b["T"] = T[:, g * Rp:(g + 1) * Rp]
            outs.append(m._b["y"])
        return y, T, outs

    def forward(self, x, Ct=None):
        """Returns the member outputs as column slices of one [M, G*N] buffer."""
        y, T, outs = self.prepare(x)
        lora = (self.A_cat if self.ln is None else self.ln_Ag, self.B_cat, self.arena.scale, T) if self.has_lora else None
        kw = {"col_scale": self.col_scale()} if self.dora else {}
        if self.ln is not None:
            kw["ln"] = (self.ln_c1, self.ln.stats_buf(), self.ln.eps, self.ln_consts if self.has_lora else None) + (self.ln._parts or (None, 0))
        self.rt.ops.gemm(x, self.W, y, lora=lora, bias=self.bias, Ct=Ct, lora_group_n=self.N if self.has_lora else 0, **kw)
        return outs

    def col_scale(self):
        return self.arena.scale_of([m.lora for m in self.members])

    def grad_slices(self, M):
        """One [M, G*N] buffer whose column slices receive the members' output gradients (attention writes them there)."""
        dy = self.buf("dy", M, self.G * self.N)
        return dy, [dy[:, g * self.N:(g + 1) * self.N] for g in range(self.G)]

    def backward(self, dy_cat, *, dres=None, out=None):
        """dx = sum_g dy_g (W_g + s B_g A_g) = dy_cat . W_cat (+ K-grouped LoRA) as ONE GEMM over K = G*N."""
        M, N, G = dy_cat.shape[0], self.N, self.G
        if getattr(self, "Wt", None) is None:
            self.Wt = self.W.t().contiguous()
            _mark_frozen(self.rt, self.Wt)
            for m in self.members:
                m.Wt = None          # the per-member transposes are dead once the stacked one exists
        dx = out if out is not None else self.buf("dx", M, self.K)
        if self.trainer is not None:
            self.trainer.linear(self.went, None, [self.members[0]._x], dy_cat)
        if not self.has_lora:
            self.rt.ops.gemm(dy_cat, self.Wt, dx, residual=dres)
            return dx
        assert self.kgrouped
        U = self.backward_operands(dy_cat)
        self.rt.ops.gemm(dy_cat, self.Wt_d if self.dora else self.Wt, dx, lora=(self.Bt_cat, self.At_cat, self.arena.scale, U), residual=dres, lora_group_k=N)
        return dx

    def backward_operands(self, dy_cat):
        """Transposed stacked weight, U buffer and (once) the members' LoRA-gradient problems for the K-grouped dX GEMM."""
        M, N, G = dy_cat.shape[0], self.N, self.G
        if getattr(self, "Wt", None) is None:
            self.Wt = self.W.t().contiguous()
            _mark_frozen(self.rt, self.Wt)
            for m in self.members:
                m.Wt = None
        Rp = self.arena.Rp
        U = self.buf("U", M, G * Rp)
        if not getattr(self, "_registered", False):
            r = self.arena.rank
            for g, m in enumerate(self.members):
                self.arena.problems += [
                    dict(P=dy_cat[:, g * N:(g + 1) * N], Q=m._b["T"], out=m.lora["gB"], M=M, Cw=N, R=r, rank_major=False),
                    dict(P=m._x, Q=U[:, g * Rp:(g + 1) * Rp], out=m.lora["gA"], M=M, Cw=self.K, R=r, rank_major=True)]
                if self.dora:
                    self.arena.dora_grads.append(dict(dY=dy_cat[:, g * N:(g + 1) * N], Y=m._y0, bias=m.bias, mag=m.lora["M"], scale=m.lora["scale"],
                                                      gmag=m.lora["gM"], gB=m.lora["gB"]))
            self._registered = True
        return U


class Conv3x3(_Module):
    """3x3 conv (pad 1) on NHWC activations as an implicit GEMM; optional stride 2 / nearest-2x upsampled input."""

    def __init__(self, rt, name, sd, arena=None, stride=1, ups=1, cin_pad=None, need_dx=True):
        super().__init__(rt, name)
        w = sd[name + ".weight"].float()               # [Cout, Cin, 3, 3]
        self.Cout, self.Cin = w.shape[0], w.shape[1]
        self.stride, self.ups = stride, ups
        self.Cin_p = cin_pad or self.Cin
        wf = torch.zeros(self.Cout, 3, 3, self.Cin_p, device=w.device)
        wf[..., : self.Cin] = w.permute(0, 2, 3, 1)
        self.Wf = wf.reshape(self.Cout, 9 * self.Cin_p).to(rt.device, rt.act).contiguous()
        self.Wb = None
        if need_dx:
            self.Cout_p = _pad_to(self.Cout, 64)
            wb = torch.zeros(self.Cin, 3, 3, self.Cout_p, device=w.device)
            wb[..., : self.Cout] = w.permute(1, 2, 3, 0)
            self.Wb = wb.reshape(self.Cin, 9 * self.Cout_p).to(rt.device, rt.act).contiguous()
        self.bias = sd[name + ".bias"].to(rt.device, F32).contiguous()
        self.lora = arena.add(name, self.Cout, 9 * self.Cin, conv_cin=self.Cin, W=self.Wf) if arena is not None else None
        self.arena = arena
        self.dora = arena is not None and arena.dora
        if self.dora:
            assert self.Cin_p == self.Cin and need_dx
            self.Wb_d = torch.empty_like(self.Wb)
            arena.dora_wts.append(dict(src=self.Wb, dst=self.Wb_d, entries=[self.lora], period=self.Cout_p, nvalid=self.Cout))
        _mark_frozen(rt, self.Wf, self.Wb)      # (DoRA's dX operand is Wb_d, rewritten every step: never marked)
        tr = rt.trainer if (rt.trainer is not None and rt.trainer.registering) else None
        self.trainer = tr
        if tr is not None:         # master weight tap-major [Cout, (ky, kx, ci)], like the forward GEMM operand
            self.went = tr.add(name + ".weight", w.permute(0, 2, 3, 1).reshape(self.Cout, 9 * self.Cin), "conv3x3")
            self.bent = tr.add(name + ".bias", sd[name + ".bias"].float(), "vector")
            tr.on_finalize(self._bind_trainer)

    def _bind_trainer(self):
        tr, Cin, Cout = self.trainer, self.Cin, self.Cout
        self.bias = tr.view(self.bent)
        for tap in range(9):       # per tap: [Cout, Cin] block of the master -> forward operand slice and (transposed) dX operand slice
            dst = self.Wf[:, tap * self.Cin_p: tap * self.Cin_p + Cin]
            dstT = self.Wb[:, tap * self.Cout_p: tap * self.Cout_p + Cout] if self.Wb is not None else None
            tr.shadow(self.went, Cout, Cin, 9 * Cin, dst, dstT, offset=tap * Cin)

    def weight_grad(self, dy):
        B, H, W, Hout, Wout = self._dims
        self.trainer.conv3x3(self.went, self.bent, self._x, dy, B=B, H=H, W=W, Cin=self.Cin, stride=self.stride, ups=self.ups)

    def geom(self, B, H, W):
        """H, W = stored input spatial dims -> (fwd ConvGeom, Hout, Wout)."""
        Hout = H * self.ups // self.stride
        Wout = W * self.ups // self.stride
        return _ops.ConvGeom(B, H, W, self.Cin_p, Hout, Wout, stride=self.stride, ups=self.ups), Hout, Wout

    def forward(self, x, B, H, W, *, rowbias=None, residual=None, key="y", out=None, train=True):
        g, Hout, Wout = self.geom(B, H, W)
        M = B * Hout * Wout
        y = out if out is not None else self.buf(key, M, self.Cout)
        lora = None
        if self.lora is not None:
            T = self.buf("T", M, self.arena.Rp) if train else None
            lora = (self.lora["A_s"], self.lora["B_s"], self.arena.scale, T)
            self._x, self._g = x, g
        if self.trainer is not None:
            self._x = x
        self._dims = (B, H, W, Hout, Wout)
        if self.dora:       # see Linear.forward
            assert rowbias is None
            y0 = y if residual is None else self.buf(key + "0", M, self.Cout)
            self.rt.ops.gemm(x, self.Wf, y0, conv=g, lora=lora, bias=self.bias, col_scale=self.lora["scale"])
            self._y0 = y0
            return y if residual is None else self.rt.ops.add2d(y0, residual, y)
        self.rt.ops.gemm(x, self.Wf, y, conv=g, lora=lora, bias=self.bias, rowbias=rowbias, rows_per_batch=Hout * Wout,
                         residual=residual)
        return y

    def backward(self, dy, *, dres=None, key="dx", out=None):
        """dy [B*Hout*Wout, Cout_p] (Cout_p = Cout rounded up to 64; only conv_out pads) -> dx [B*H*W, Cin] (+ dres)."""
        B, H, W, Hout, Wout = self._dims
        rt = self.rt
        assert dy.shape[1] == self.Cout_p, "conv backward wants dy padded to a multiple of 64 channels"
        if self.trainer is not None:
            self.weight_grad(dy)
        if self.ups == 2:
            # dX of conv(up2(x)): transposed conv at the upsampled resolution, then 2x2 block sums
            gb = _ops.ConvGeom(B, Hout, Wout, self.Cout_p, Hout, Wout, flip=1)
            dup = self.buf("dup", B * Hout * Wout, self.Cin)
            rt.ops.gemm(dy, self.Wb, dup, conv=gb)
            dx = out if out is not None else self.buf(key, B * H * W, self.Cin)
            assert dres is None
            return rt.ops.sum2x2(dup, dx, B=B, H=H, W=W)
        gb = _ops.ConvGeom(B, Hout, Wout, self.Cout_p, H, W, flip=1, tr=int(self.stride == 2))
        dx = out if out is not None else self.buf(key, B * H * W, self.Cin)
        if self.lora is None:
            return rt.ops.gemm(dy, self.Wb, dx, conv=gb, residual=dres)
        # LoRA conv (stride 1): U = s * dy . Bup (per pixel);  dx = convT(dy, W) + convT(U, A) (+ dres)
        M = B * Hout * Wout
        U64 = self.buf("U64", M, 64, zero=True)
        U = U64[:, : self.arena.Rp]
        rt.ops.gemm(dy, self.lora["Bt_s"], U, alpha=self.arena.scale)
        rt.ops.gemm(dy, self.Wb_d if self.dora else self.Wb, dx, conv=gb, residual=dres)
        rt.ops.gemm(U64, self.lora["Ab_s"], dx, conv=_ops.ConvGeom(B, Hout, Wout, 64, H, W, flip=1), residual=dx)
        if not getattr(self, "_registered", False):
            r = self.arena.rank
            self.arena.problems += [
                dict(P=dy, Q=self._b["T"], out=self.lora["gB"], M=M, Cw=self.Cout, R=r, rank_major=False),
                dict(P=self._x, Q=U, out=self.lora["gA"], M=M, Cw=9 * self.Cin, R=r, rank_major=True, conv=self._g)]
            if self.dora:
                self.arena.dora_grads.append(dict(dY=dy, Y=self._y0, bias=self.bias, mag=self.lora["M"], scale=self.lora["scale"],
                                                  gmag=self.lora["gM"], gB=self.lora["gB"]))
            self._registered = True
        return dx


def _register_affine(layer, rt, name, sd):
    """full fine-tune: gamma / beta are read by the norm kernels in fp32 straight from the trainer's master arena."""
    tr = rt.trainer if (rt.trainer is not None and rt.trainer.registering) else None
    layer.trainer = tr
    if tr is None:
        return
    layer.gent = tr.add(name + ".weight", sd[name + ".weight"].float(), "vector")
    layer.bent = tr.add(name + ".bias", sd[name + ".bias"].float(), "vector")

    def bind():
        layer.gamma, layer.beta = tr.view(layer.gent), tr.view(layer.bent)
    tr.on_finalize(bind)


class GroupNorm(_Module):
    def __init__(self, rt, name, sd, eps, silu):
        super().__init__(rt, name)
        self.gamma = sd[name + ".weight"].to(rt.device, F32).contiguous()
        self.beta = sd[name + ".bias"].to(rt.device, F32).contiguous()
        self.eps, self.silu, self.C = eps, silu, self.gamma.numel()
        _register_affine(self, rt, name, sd)

    def forward(self, x1, x2, B, HW):
        y = self.buf("y", B * HW, self.C)
        if "stats" not in self._b:
            self._b["stats"] = self.rt.gn_stats("fwd", B * 64)
        stats = self._b["stats"]
        self._in = (x1, x2, B, HW)
        return self.rt.ops.groupnorm_fwd(x1, x2, y, stats, B=B, HW=HW, gamma=self.gamma, beta=self.beta, eps=self.eps, silu=self.silu,
                                         stats_zeroed=self.rt.gn_prezero)

    def backward(self, dy, dres=None, out=None, colsum_ws=None):
        """colsum_ws: see ops.groupnorm_bwd (the per-image column sums of dx as a side output of the apply kernel)."""
        x1, x2, B, HW = self._in
        dx = out if out is not None else self.buf("dx", B * HW, self.C)
        if self.trainer is not None:
            self.trainer.affine(groupnorm=True, x1=x1, x2=x2, dy=dy, stats=self._b["stats"], gamma=self.gamma, beta=self.beta, gent=self.gent,
                                bent=self.bent, B=B, HW=HW, eps=self.eps, silu=self.silu)
        if "bstats" not in self._b:
            self._b["bstats"] = self.rt.gn_stats("bwd", B * 64)
        return self.rt.ops.groupnorm_bwd(x1, x2, dy, dx, self._b["stats"], self._b["bstats"], B=B, HW=HW,
                                         gamma=self.gamma, beta=self.beta, eps=self.eps, silu=self.silu, dres=dres, stats_zeroed=self.rt.gn_prezero,
                                         **({"colsum_ws": colsum_ws} if colsum_ws is not None else {}))


# SDLT_LN_NOOP (bit 0: forward, bit 1: backward): a TIMING-ONLY switch that skips the UNet's LayerNorm launches - the upper bound of what folding
# them into the neighbouring GEMMs could gain, measured before building the fold.  The results are garbage.
_LN_NOOP = int(os.environ.get("SDLT_LN_NOOP", "0"))


COLSUM_FUSED = os.environ.get("SDLT_COLSUM_FUSED", "1") != "0"      # time-embedding column sums as a side output of norm2's backward (A/B switch)
LN_FOLD = int(os.environ.get("SDLT_LN_FOLD", "7"))      # bit 0: norm1 -> to_q|to_k|to_v, bit 1: norm2 -> attn2.to_q, bit 2: norm3 -> ff.net.0.proj (A/B switch)


LN_FOLD_WIDTH = int(os.environ.get("SDLT_LN_FOLD_WIDTH", "1280"))   # fold in blocks whose width is a multiple of this (1280: where the producers leave row partials at batch 1)
PARTS = os.environ.get("SDLT_LN_PARTS", "1") != "0"    # row partials from the producing GEMM (A/B switch; 0: every folded consumer computes its statistics)


class LayerNorm(_Module):
    def __init__(self, rt, name, sd, eps=1e-5):
        super().__init__(rt, name)
        self.gamma = sd[name + ".weight"].to(rt.device, F32).contiguous()
        self.beta = sd[name + ".bias"].to(rt.device, F32).contiguous()
        self.eps = eps
        _register_affine(self, rt, name, sd)

    folded = False      # the forward runs inside the consumer GEMM (Linear.fold_ln / StackedLinear.fold_ln)
    _parts = None       # (fp32 [M, P, 2], P): row partials the producer of this pass's input left (Linear.forward parts_for=), or None
    emit_y = False      # ... and the consumer has an adapter: the backward also writes the normalised rows for the adapter-gradient launch

    def stats_buf(self):
        return self.buf("stats", self._x.shape[0] * 2, dtype=F32)

    def ybuf(self):
        return self.buf("yln", *self._x.shape)

    def forward(self, x, out=None, unfolded=False):
        self._x = x
        if self.folded and not unfolded:           # no launch: the consumer reads the raw rows and writes (mean, rstd) into stats_buf()
            assert out is None
            return x
        y = out if out is not None else self.buf("y", *x.shape)
        if _LN_NOOP & 1:        # TIMING EXPERIMENT ONLY (SDLT_LN_NOOP): the launch is skipped, y keeps stale data - never set in a real run
            self.buf("stats", x.shape[0] * 2, dtype=F32)
            return y
        return self.rt.ops.layernorm_fwd(x, y, self.buf("stats", x.shape[0] * 2, dtype=F32), gamma=self.gamma, beta=self.beta, eps=self.eps)

    def backward(self, dy, dres=None, out=None):
        dx = out if out is not None else self.buf("dx", *self._x.shape)
        if self.trainer is not None:
            self.trainer.affine(groupnorm=False, x1=self._x, x2=None, dy=dy, stats=self._b["stats"], gamma=self.gamma, beta=self.beta,
                                gent=self.gent, bent=self.bent, B=1, HW=self._x.shape[0])
        if _LN_NOOP & 2:
            return dx
        if self.folded and self.emit_y:
            return self.rt.ops.layernorm_bwd(self._x, dy, dx, self._b["stats"], gamma=self.gamma, dres=dres, beta=self.beta, y_out=self.ybuf())
        return self.rt.ops.layernorm_bwd(self._x, dy, dx, self._b["stats"], gamma=self.gamma, dres=dres)


# ---------------------------------------------------------------------------------------- composite blocks

CTX_PAD = 128  # the 77 text tokens are stored as 128 rows per batch: 16-byte aligned transposed tiles, a whole number of
               # 64-wide K steps for the DAAM score-gradient GEMMs, and the row count the CLIP plan writes (clip.TP)


class Attention(_Module):
    def __init__(self, rt, name, sd, arena, heads, cross, hooked):
        super().__init__(rt, name)
        self.to_q = Linear(rt, name + ".to_q", sd, arena)
        self.to_k = Linear(rt, name + ".to_k", sd, arena)
        self.to_v = Linear(rt, name + ".to_v", sd, arena)
        self.to_out = Linear(rt, name + ".to_out.0", sd, arena)
        self.heads, self.cross, self.hooked = heads, cross, hooked
        # one GEMM for the projections that share an input: q|k|v of self-attention, k|v of cross-attention
        self.stack = StackedLinear(rt, name + (".to_kv" if cross else ".to_qkv"), [self.to_k, self.to_v] if cross else [self.to_q, self.to_k, self.to_v])
        self.kv_batched = False
        self.daam_batched = False
        self.C = self.to_q.N
        self.d = self.C // heads
        self.scale = 1.0 / math.sqrt(self.d)

    def forward(self, x, ctx, B, N, residual, parts_for=None):
        """x [B*N, C]; ctx [B*CTX_PAD, D] for cross attention.  Returns residual + to_out(attn)."""
        rt, C = self.rt, self.C
        kv = ctx if self.cross else x
        Nk, Nkp = (77, CTX_PAD) if self.cross else (N, N)
        Mq, Mk = B * N, B * Nkp
        # transposed copies of Q and K are only needed as GEMM operands of the score side output's backward (hooked
        # cross-attention); the attention kernels themselves read every tile in its natural layout
        need_t = self.cross and self.hooked
        if self.cross:
            q = self.to_q.forward(x, Ct=self.buf("Qt", C, Mq) if need_t else None)
            if self.kv_batched:          # to_k|to_v of every cross-attention layer ran in one batched launch (UNet._cross_kv_forward)
                k, v = self.to_k._b["y"], self.to_v._b["y"]
            else:
                KVt = self.buf("KVt", 2 * C, _pad_to(Mk, 8)) if need_t else None
                k, v = self.stack.forward(kv, Ct=KVt)
                if need_t:
                    self._b["Kt"] = KVt[:C]
        else:
            q, k, v = self.stack.forward(x)
        O = self.buf("O", Mq, C)
        L = self.buf("L", B * self.heads * N, dtype=F32)
        self._dims = (B, N, Nk, Nkp)
        # self-attention whose to_out.0 input gradient runs on the wave-split-K kernel: that product leaves the backward's row term D = rowsum(dO o O) as a side output
        # (no D pre-pass launch); its slots are cleared here, by the forward kernel's epilogue
        self._rowdot = (not self.cross and getattr(self, "_rowdot_ok", True) and self.to_out.trainer is None
                        and getattr(rt.ops, "wsk_rowdot_shape", None) is not None and rt.ops.wsk_rowdot_shape(Mq, C, C, self.to_out.lora is not None, self.d))
        kwz = {"zero_D": self.buf("D", B * self.heads * N, dtype=F32)} if self._rowdot else {}
        rt.ops.attn_fwd(q, k, v, None, O, L, B=B, H=self.heads, Nq=N, Nk=Nk, Nqp=N, Nkp=Nkp, d=self.d, scale=self.scale, **kwz)
        if self.cross and self.hooked:
            # DAAM side output (ti_cross_attn_loss.py:201-212): sum over heads of Q_h K_h^T / sqrt(d) = Q K^T / sqrt(d).
            # The token-attention loss only ever uses the MEAN over layers of these maps (loss.py:23-52), so the layers
            # of one resolution accumulate into one fp32 sum (GEMM epilogue, C += ...).
            if N not in rt.daam_sums:
                rt.daam_sums[N] = [rt.zeros(B * N, CTX_PAD, dtype=F32), 0, False]
            ent = rt.daam_sums[N]
            if rt.defer_daam_scores:      # all hooked layers of a resolution in one batched launch after the forward (UNet._daam_forward)
                self._qk = (q, k)
            else:
                for b in range(B):
                    rt.ops.gemm(q[b * N:(b + 1) * N], k[b * Nkp:(b + 1) * Nkp], ent[0][b * N:(b + 1) * N], alpha=self.scale,
                                accumulate=ent[2])
            ent[2] = True
            ent[1] += 1
            if rt.keep_daam_maps:
                S = self.buf("S", B * N, CTX_PAD, dtype=F32)
                for b in range(B):
                    rt.ops.gemm(q[b * N:(b + 1) * N], k[b * Nkp:(b + 1) * Nkp], S[b * N:(b + 1) * N], alpha=self.scale)
                rt.daam.append((self.name, S.view(B, N, CTX_PAD)))
        return self.to_out.forward(O, residual=residual, parts_for=parts_for)

    def backward(self, dout, dctx=None):
        """dout = grad of (residual + to_out(attn)); returns d(attention input) WITHOUT the residual path."""
        rt, C = self.rt, self.C
        B, N, Nk, Nkp = self._dims
        Mq, Mk = B * N, B * Nkp
        D = self.buf("D", B * self.heads * N, dtype=F32)
        rd = dict(O=self._b["O"], D=D, Nq=N, done=False) if getattr(self, "_rowdot", False) else None
        dO = self.to_out.backward(dout, **({"rowdot": rd} if rd is not None else {}))
        if rd is not None and not rd["done"]:
            self._rowdot_ok = False      # (this product does not run where the side output exists: the next forward stops clearing D)
        q, k, v = self.to_q._b["y"], self.to_k._b["y"], self.to_v._b["y"]
        fused = self.stack.has_lora and self.stack.kgrouped or not self.stack.has_lora
        if fused and self.cross:
            dq = self.buf("dq", Mq, C)
            dkv, (dk, dv) = self.stack.grad_slices(Mk)
        elif fused:
            dqkv, (dq, dk, dv) = self.stack.grad_slices(Mq)
        else:
            dq, dk, dv = self.buf("dq", Mq, C), self.buf("dk", Mk, C), self.buf("dv", Mk, C)
        kw = {"d_ready": True} if (rd is not None and rd["done"]) else {}
        if self.cross:
            # cross-attention: ~160 workgroups, each owning all 77 keys of one head and a range of query tiles; the fp32
            # dK/dV accumulators are adjacent so the kernel side zeroes / converts them with one launch each
            qs = max(2, min((N + 63) // 64, XATTN_WGS // max(1, self.heads * B)))
            lay_ = rt.daam_layer_grads.get(self.name) if (self.hooked and rt.daam_layer_grads) else None
            pre_ = self.hooked and rt.daam_grads is not None and rt.daam_applied and self.daam_batched and lay_ is None
            # (a hooked layer whose score-gradient GEMMs still run per layer reads dk right after this kernel: no deferral then)
            self._defer_sum = self.kv_batched and fused and self.d <= 96 and CTX_PAD <= 128 and lay_ is None and (pre_ or not (self.hooked and rt.daam_grads is not None))
            if self._defer_sum:      # the slabs of ALL layers are summed by one launch in UNet._cross_kv_backward: layer-owned, not scratch
                kv32 = self.buf("dkv32", 2 * qs * Mk, C, dtype=F32)
            else:
                kv32 = rt.scratch("attn_dkv32", 2 * qs * Mk * C).view(2 * qs * Mk, C)    # partial slabs, consumed inside attn_bwd
            kw = dict(qsplit=qs, dK32=kv32[:qs * Mk], dV32=kv32[qs * Mk:])
            if self._defer_sum:
                kw["defer_splitsum"] = True
        lay = rt.daam_layer_grads.get(self.name) if (self.cross and self.hooked and rt.daam_layer_grads) else None
        pre = self.cross and self.hooked and rt.daam_grads is not None and rt.daam_applied and self.daam_batched and lay is None
        if pre:     # dq / dk already hold the score side output's gradient (UNet.daam_backward, one batched GEMM per group)
            kw.update(accumulate_dq=True, accumulate_dk=True)
        rt.ops.attn_bwd(q, k, v, None, None, self._b["O"], self._b["L"], dO, None, D, dq, dk, dv,
                        B=B, H=self.heads, Nq=N, Nk=Nk, Nqp=N, Nkp=Nkp, d=self.d, scale=self.scale, **kw)
        if self.cross and getattr(self, "_defer_sum", False):
            self._sum_item = dict(dK32=kw["dK32"], dV32=kw["dV32"], dK=dk, dV=dv, nsplit=kw["qsplit"], B=B, Nk=Nk, Nkp=Nkp, acc0=bool(kw.get("accumulate_dk")))
        if self.cross and self.hooked and (lay is not None or (rt.daam_grads is not None and not pre)):
            # backward of the score side output: S = a Q K^T  ->  dQ += a dS K,  dK += a dS^T Q   (shared dS per resolution, or this
            # layer's own when the maps are autograd outputs of the call-compatible module, shim._UNetFn)
            dS, dSt = lay if lay is not None else rt.daam_grads[N]
            Kt, Qt = self._b["Kt"], self._b["Qt"]
            for b in range(B):
                rt.ops.gemm(dS[b * N:(b + 1) * N], Kt[:, b * Nkp:(b + 1) * Nkp], dq[b * N:(b + 1) * N], residual=dq[b * N:(b + 1) * N],
                            alpha=self.scale)
                rt.ops.gemm(dSt[b * Nkp:(b + 1) * Nkp], Qt[:, b * N:(b + 1) * N], dk[b * Nkp:(b + 1) * Nkp], residual=dk[b * Nkp:(b + 1) * Nkp],
                            alpha=self.scale)
        if fused and self.cross:
            dx = self.to_q.backward(dq)
            if not self.kv_batched:   # (batched: all layers' dkv go through one launch at the end of UNet.backward)
                self.stack.backward(dkv, dres=dctx, out=dctx)      # gradient w.r.t. the text conditioning, accumulated over every cross-attention layer
        elif fused:
            dx = self.stack.backward(dqkv)
        else:
            dx = self.to_q.backward(dq)
            if self.cross:
                self.to_k.backward(dk, dres=dctx, out=dctx)
                self.to_v.backward(dv, dres=dctx, out=dctx)
            else:
                self.to_k.backward(dk, dres=dx, out=dx)
                self.to_v.backward(dv, dres=dx, out=dx)
        return dx


class TransformerBlock(_Module):
    def __init__(self, rt, name, sd, arena, heads, hooked):
        super().__init__(rt, name)
        self.norm1 = LayerNorm(rt, name + ".norm1", sd)
        self.attn1 = Attention(rt, name + ".attn1", sd, arena, heads, cross=False, hooked=False)
        self.norm2 = LayerNorm(rt, name + ".norm2", sd)
        self.attn2 = Attention(rt, name + ".attn2", sd, arena, heads, cross=True, hooked=hooked)
        self.norm3 = LayerNorm(rt, name + ".norm3", sd)
        self.ff1 = Linear(rt, name + ".ff.net.0.proj", sd)
        self.ff2 = Linear(rt, name + ".ff.net.2", sd)
        # GEGLU fused into the GEMMs on either side (sdlt_gemm_params.epi_op): ff.net.0.proj's rows are permuted into the
        # interleaved-16 layout so that a hidden column and its gate share an output tile; hidden * gelu(gate) leaves the forward
        # GEMM's epilogue and GEGLU's backward is the epilogue of ff.net.2's dX GEMM - two launches fewer per block and direction.
        # (Not with the full fine-tune: its master weights / weight gradients keep the checkpoint's row order.)
        H = self.ff2.K
        # Round 2 measured the fused epilogues as a net LOSS (49.65 ms unfused vs 50.6 ms fused) and left them off: the library erff inside
        # a GEMM epilogue is ~40 serial VALU instructions per element.  With the one-exponential GELU of round 3 (common.h) they pay: SDXL 46.52 ms
        # unfused, 46.10 fused at C >= 1280, 45.96 everywhere; SD1.5 25.59 -> 25.30.  Default: fused; SDLT_GEGLU_MIN_C=<width> restricts them to
        # transformer widths >= the value (a huge value = the element-wise kernels).
        self.fused_geglu = (self.ff1.trainer is None and H % 16 == 0 and hasattr(rt.ops, "geglu_perm")
                            and self.ff2.N >= int(os.environ.get("SDLT_GEGLU_MIN_C", "0")))
        perm = None
        if self.fused_geglu:
            perm = rt.ops.geglu_perm(H, self.ff1.W.device)
            self.ff1.W = self.ff1.W[perm].contiguous()
            self.ff1.Wt = self.ff1.W.t().contiguous()
            _mark_frozen(rt, self.ff1.W, self.ff1.Wt)
            self.ff1.bias = self.ff1.bias[perm].contiguous()
        # LayerNorm forward folded into the GEMM behind it (sdlt_gemm_params.ln_c1): norm1 -> to_q|to_k|to_v, norm2 -> attn2.to_q,
        # norm3 -> ff.net.0.proj (with the fused GEGLU epilogue only).  Not with the full fine-tune (gamma / beta / W are trained), DoRA (the
        # column factor needs the unfolded rows); rank pads above 16 fold norm3 only (round 6: ff.net.0.proj carries no adapter - the folded ADAPTER products exist for
        # rank pad 16).  SDLT_LN_FOLD=0: the LayerNorm launches (bit mask per norm).
        if (LN_FOLD and hasattr(rt.ops, "LnFoldPlan") and self.norm1.trainer is None and arena is not None and self.attn1.C % LN_FOLD_WIDTH == 0):
            w = lambda n: sd[n + ".weight"].float()
            a1 = self.attn1
            ok = lambda l: l.trainer is None and not l.dora        # (the folded kernel variants exist for rank-16 adapter products and for ff.net.0.proj + GEGLU)
            if (LN_FOLD & 1) and arena.Rp == 16 and ok(a1.to_q):
                a1.stack.fold_ln(self.norm1, [w(m.name) for m in a1.stack.members])
            if (LN_FOLD & 2) and arena.Rp == 16 and ok(self.attn2.to_q):
                self.attn2.to_q.fold_ln(self.norm2, w(self.attn2.to_q.name))
            if self.fused_geglu and (LN_FOLD & 4) and ok(self.ff1):      # (no adapter on ff.net.0.proj: any rank, DoRA too - round 6)
                w3 = w(self.ff1.name).to(rt.device)
                self.ff1.fold_ln(self.norm3, w3[perm], keep_plain=True)

    def forward(self, x, ctx, B, N, next_norm=None):
        """next_norm: the LayerNorm that reads this block's output (the next block's norm1) - see Linear.forward parts_for."""
        x1 = self.attn1.forward(self.norm1.forward(x), None, B, N, residual=x, parts_for=self.norm2)
        x2 = self.attn2.forward(self.norm2.forward(x1), ctx, B, N, residual=x1, parts_for=self.norm3)
        if self.fused_geglu:
            g = self.buf("g", x.shape[0], self.ff2.K)
            # norm3's fold pays only with the producer's row partials (K-walk statistics cost ff.net.0.proj more than the LayerNorm launch: DESIGN 4.12):
            # without them - other batch sizes / resolutions than 1024 rows, concurrent-job hint - this pass runs the launch and the unfolded operand
            plain = self.ff1.ln is not None and self.norm3._parts is None
            self.ff1.forward(self.norm3.forward(x2, unfolded=plain), geglu_out=g, unfolded=plain)
        else:
            f1 = self.ff1.forward(self.norm3.forward(x2))
            g = self.rt.ops.geglu_fwd(f1, self.buf("g", x.shape[0], f1.shape[1] // 2))
        return self.ff2.forward(g, residual=x2, parts_for=next_norm)

    def backward(self, dx3, dctx):
        rt = self.rt
        if self.fused_geglu:
            f1 = self.ff1._b["y"]
            df1 = rt.ops.gemm(dx3, self.ff2.Wt, None, geglu_bwd=(f1, self.buf("df1", *f1.shape)))
        else:
            dg = self.ff2.backward(dx3)
            df1 = rt.ops.geglu_bwd(self.ff1._b["y"], dg, self.buf("df1", *self.ff1._b["y"].shape))
        dx2 = self.norm3.backward(self.ff1.backward(df1), dres=dx3)
        dx1 = self.norm2.backward(self.attn2.backward(dx2, dctx), dres=dx2)
        return self.norm1.backward(self.attn1.backward(dx1), dres=dx1)


class Transformer2D(_Module):
    def __init__(self, rt, name, sd, arena, C, heads, nlayers, hooked):
        super().__init__(rt, name)
        self.norm = GroupNorm(rt, name + ".norm", sd, 1e-6, silu=False)
        self.proj_in = Linear(rt, name + ".proj_in", sd)
        self.blocks = [TransformerBlock(rt, f"{name}.transformer_blocks.{k}", sd, arena, heads, hooked) for k in range(nlayers)]
        self.proj_out = Linear(rt, name + ".proj_out", sd)

    def forward(self, x, ctx, B, HW):
        h = self.proj_in.forward(self.norm.forward(x, None, B, HW))
        self.blocks[0].norm1._parts = None            # (proj_in leaves no row partials)
        for i, blk in enumerate(self.blocks):
            h = blk.forward(h, ctx, B, HW, next_norm=self.blocks[i + 1].norm1 if i + 1 < len(self.blocks) else None)
        return self.proj_out.forward(h, residual=x)

    def backward(self, dout, dctx):
        dh = self.proj_out.backward(dout)
        for blk in reversed(self.blocks):
            dh = blk.backward(dh, dctx)
        return self.norm.backward(self.proj_in.backward(dh), dres=dout)


class ResnetBlock(_Module):
    def __init__(self, rt, name, sd, arena, cin1, cin2, cout):
        super().__init__(rt, name)
        self.cin1, self.cin2, self.cout = cin1, cin2, cout
        self.norm1 = GroupNorm(rt, name + ".norm1", sd, 1e-5, silu=True)
        self.conv1 = Conv3x3(rt, name + ".conv1", sd)
        self.temb = Linear(rt, name + ".time_emb_proj", sd)
        self.norm2 = GroupNorm(rt, name + ".norm2", sd, 1e-5, silu=True)
        self.conv2 = Conv3x3(rt, name + ".conv2", sd, arena)
        self.shortcut = None
        if (name + ".conv_shortcut.weight") in sd:
            self.shortcut = Linear(rt, name + ".conv_shortcut", sd)
            if cin2:
                w = self.shortcut
                w.W1, w.W2 = w.W[:, :cin1], w.W[:, cin1:]

    def forward(self, x1, x2, semb, B, H, W):
        """x1 (+x2: skip tensor, channel-concatenated) -> out [B*H*W, cout]; semb = silu(time embedding) [B, tdim]."""
        rt = self.rt
        HW = H * W
        h = self.norm1.forward(x1, x2, B, HW)
        tp = self._tp if getattr(self, "_tp", None) is not None else self.temb.forward(semb, train=False)     # (stacked: UNet._forward)
        c1 = self.conv1.forward(h, B, H, W, rowbias=tp)
        h2 = self.norm2.forward(c1, None, B, HW)
        if self.shortcut is None:
            sc = x1
        else:
            s = self.shortcut
            sc = self.buf("sc", B * HW, self.cout)
            if x2 is None:
                rt.ops.gemm(x1, s.W, sc, bias=s.bias)
            else:
                rt.ops.gemm(x1, s.W1, sc, X2=x2, W2=s.W2, bias=s.bias)
        self._in = (x1, x2, B, H, W)
        return self.conv2.forward(h2, B, H, W, residual=sc)

    def backward(self, dout):
        """dout [M, cout] -> dx [M, cin1 + cin2] (caller splits the concat)."""
        rt = self.rt
        x1, x2, B, H, W = self._in
        dh2 = self.conv2.backward(dout)
        # h = conv1(.) + time_emb_proj(silu(emb))[b]: d(proj output)[b] = column sums of dc1 over the pixels of image b, accumulated over all resnets into
        # d silu(emb).  Stacked (batch 1): norm2's backward leaves the partial sums as a side output, ONE launch finishes them for all resnets and one GEMM
        # applies the projection at the end of UNet._backward (were a column-sum launch and its finish per resnet)
        slot = getattr(self, "_dtp_slot", None) if (rt.want_dpooled or self.temb.trainer is not None) else None
        fused = slot is not None and COLSUM_FUSED
        dc1 = self.norm2.backward(dh2, colsum_ws=self._dtp_ws if fused else None)
        if (rt.want_dpooled or self.temb.trainer is not None) and not fused:
            if slot is not None:
                rt.ops.colsum(dc1, slot, B=B, R=H * W)
            else:
                dtp = rt.ops.colsum(dc1, self.buf("dtp", B, self.cout), B=B, R=H * W)
                self.temb.backward(dtp, dres=rt.dsemb, out=rt.dsemb)
        dh1 = self.conv1.backward(dc1)
        if self.shortcut is not None and self.shortcut.trainer is not None:      # its forward is issued here, not through Linear.forward
            self.shortcut._x = x1
            self.shortcut.trainer = None                                       # (the hook inside backward would not know about x2)
            try:
                dres = self.shortcut.backward(dout, key="dsc")
            finally:
                self.shortcut.trainer = rt.trainer
            self.shortcut.weight_grad(dout, xs=[x1] if x2 is None else [x1, x2])
        else:
            dres = dout if self.shortcut is None else self.shortcut.backward(dout, key="dsc")
        return self.norm1.backward(dh1, dres=dres)


class UNet(_Module):
    """forward(noisy NHWC, timesteps, ctx[, pooled, time_ids]) -> eps_hat [B*h*w, 4] fp32;  backward(dpred)."""

    def __init__(self, rt, version, sd, lora_rank=None, lora_alpha_multiplier=1.0, trainer=None, use_dora=False):
        """trainer: a fullft.WeightTrainer -> every parameter of the UNet is registered with it and trained (the
        reference's `is_lora = False` branch, main.py:144-149); lora_rank must then be None."""
        super().__init__(rt, "unet")
        cfg = CONFIGS[version] if isinstance(version, str) else version
        self.cfg = cfg
        self.trainer = trainer
        if trainer is not None:
            assert lora_rank is None, "full fine-tune and LoRA are exclusive (config.is_lora)"
            rt.trainer = trainer
            trainer.registering = True
        boc = cfg["block_out_channels"]
        for c in boc:
            assert c % 64 == 0, "channel counts must be multiples of 64 (GEMM K-step / GroupNorm tiling)"
        self.arena = LoraArena(rt, lora_rank, lora_alpha_multiplier, dora=use_dora) if lora_rank else None
        ar, L = self.arena, cfg["layers_per_block"]
        c0 = boc[0]
        self.tdim = c0 * TIME_DIM_MULT
        self.conv_in = Conv3x3(rt, "conv_in", sd, cin_pad=64, need_dx=False)
        self.t1 = Linear(rt, "time_embedding.linear_1", sd, need_dx=False)
        self.t2 = Linear(rt, "time_embedding.linear_2", sd, need_dx=trainer is not None)
        if cfg["addition"]:
            self.a1 = Linear(rt, "add_embedding.linear_1", sd)
            self.a2 = Linear(rt, "add_embedding.linear_2", sd)
        self.down, self.up = [], []
        out_c = c0
        for i, c in enumerate(boc):
            in_c, out_c = out_c, c
            res, att = [], []
            for j in range(L):
                res.append(ResnetBlock(rt, f"down_blocks.{i}.resnets.{j}", sd, ar, in_c if j == 0 else out_c, 0, out_c))
                if cfg["down_has_attn"][i]:
                    att.append(Transformer2D(rt, f"down_blocks.{i}.attentions.{j}", sd, ar, out_c, cfg["heads"][i],
                                             cfg["transformer_layers"][i], True))
            ds = Conv3x3(rt, f"down_blocks.{i}.downsamplers.0.conv", sd, stride=2) if i != len(boc) - 1 else None
            self.down.append((res, att, ds))
        cm = boc[-1]
        self.mid = (ResnetBlock(rt, "mid_block.resnets.0", sd, ar, cm, 0, cm),
                    Transformer2D(rt, "mid_block.attentions.0", sd, ar, cm, cfg["heads"][-1], cfg["transformer_layers"][-1], False),
                    ResnetBlock(rt, "mid_block.resnets.1", sd, ar, cm, 0, cm))
        rev, rev_h, rev_l = list(reversed(boc)), list(reversed(cfg["heads"])), list(reversed(cfg["transformer_layers"]))
        out_c = rev[0]
        for i in range(len(boc)):
            prev, out_c = out_c, rev[i]
            in_c = rev[min(i + 1, len(boc) - 1)]
            res, att = [], []
            for j in range(L + 1):
                skip = in_c if j == L else out_c
                rin = prev if j == 0 else out_c
                res.append(ResnetBlock(rt, f"up_blocks.{i}.resnets.{j}", sd, ar, rin, skip, out_c))
                if cfg["up_has_attn"][i]:
                    att.append(Transformer2D(rt, f"up_blocks.{i}.attentions.{j}", sd, ar, out_c, rev_h[i], rev_l[i], True))
            us = Conv3x3(rt, f"up_blocks.{i}.upsamplers.0.conv", sd, ups=2) if i != len(boc) - 1 else None
            self.up.append((res, att, us))
        self.norm_out = GroupNorm(rt, "conv_norm_out", sd, 1e-5, silu=True)
        self.conv_out = Conv3x3(rt, "conv_out", sd)
        if ar is not None:
            ar.finalize()
        if trainer is not None:
            trainer.registering = False
            trainer.finalize()
        self._grad_plan = None
        # time_emb_proj of every resnet reads the same silu(emb): ONE GEMM over the stacked weights [sum Cout, tdim] per pass instead of
        # one M = B launch per resnet (17 in SDXL, 22 in SD1.5); the same for their input gradients (dsemb) when B == 1.  Not with the
        # full fine-tune (the projections are trained layer by layer there).
        self.resnets = [r for (res, _, _) in self.down for r in res] + [self.mid[0], self.mid[2]] + [r for (res, _, _) in self.up for r in res]
        self.temb_W = None
        if trainer is None:
            self.temb_W = torch.cat([r.temb.W for r in self.resnets], 0).contiguous()
            self.temb_b = torch.cat([r.temb.bias for r in self.resnets]).contiguous()
            self.temb_Wt = self.temb_W.t().contiguous()
            off = 0
            for r in self.resnets:                  # the members keep working on views of the stacked operands (B > 1 backward)
                r.temb_off = off
                r.temb.W, r.temb.Wt = self.temb_W[off: off + r.cout], self.temb_Wt[:, off: off + r.cout]
                off += r.cout
        # cross-attention to_k|to_v of every layer read the same text conditioning: one batched launch per (width, hooked)
        # group in forward, and one for all their input gradients in backward (instead of 2 x 70 M = 128 GEMMs)
        tfs = [t for (_, att, _) in self.down for t in att] + [self.mid[1]] + [t for (_, att, _) in self.up for t in att]
        self.cross_attns = [blk.attn2 for t in tfs for blk in t.blocks]
        ok = all(a.stack.has_lora and a.stack.kgrouped for a in self.cross_attns) and len(self.cross_attns) > 1
        self._kv_groups = None
        for a in self.cross_attns:
            a.kv_batched = ok

    # ------------------------------------------------------------------------------------ forward
    def forward(self, x, timesteps_f, ctx, pooled=None, time_ids=None, *, B, H, W):
        """x: noisy latent NHWC padded to 64 channels [B*H*W, 64]; timesteps_f fp32 [B]; ctx [B*CTX_PAD, D];
        pooled [B, P] (act dtype), time_ids fp32 [B*6] (SDXL).  Returns eps_hat fp32 [B*H*W, 4]."""
        rt, cfg = self.rt, self.cfg
        rt.gn_clear("fwd")          # one fill for the statistics of every GroupNorm of this pass
        rt.gn_prezero = True
        try:
            return self._forward(x, timesteps_f, ctx, pooled, time_ids, B=B, H=H, W=W)
        finally:
            rt.gn_prezero = False

    def _forward(self, x, timesteps_f, ctx, pooled=None, time_ids=None, *, B, H, W):
        rt, cfg = self.rt, self.cfg
        rt.daam = []
        for ent in rt.daam_sums.values():
            ent[1], ent[2] = 0, False
        boc = cfg["block_out_channels"]
        te = rt.ops.timestep_embedding(timesteps_f, self.buf("te", B, boc[0]))
        e1 = self.t1.forward(te, train=False)
        emb = self.t2.forward(rt.ops.map_bf16(_ops.MAP_SILU, e1, None, self.buf("e1s", *e1.shape)), train=False)
        if cfg["addition"]:
            tdim_add = cfg["addition_time_embed_dim"]
            P = cfg["proj_class_in"] - 6 * tdim_add
            add_in = self.buf("add_in", B, cfg["proj_class_in"])
            add_in[:, :P].copy_(pooled)
            for b in range(B):   # sinusoids of the 6 SDXL time ids land next to the pooled text embedding
                rt.ops.timestep_embedding(time_ids[b * 6:(b + 1) * 6], add_in[b, P:].view(6, tdim_add))
            a1 = self.a1.forward(add_in, train=False)
            emb = self.a2.forward(rt.ops.map_bf16(_ops.MAP_SILU, a1, None, self.buf("a1s", *a1.shape)), residual=emb, train=False)
        semb = rt.ops.map_bf16(_ops.MAP_SILU, emb, None, self.buf("semb", *emb.shape))
        self._b["semb_in"] = emb
        if self.temb_W is not None:
            tp_all = rt.ops.gemm(semb, self.temb_W, self.buf("tp_all", B, self.temb_W.shape[0]), bias=self.temb_b)
            for r in self.resnets:
                r._tp = tp_all[:, r.temb_off: r.temb_off + r.cout]

        self._cross_kv_forward(ctx, B)
        h = self.conv_in.forward(x, B, H, W, train=False)
        skips = [(h, H, W)]
        ch, cw = H, W
        for (res, att, ds) in self.down:
            for j, r in enumerate(res):
                h = r.forward(h, None, semb, B, ch, cw)
                if att:
                    h = att[j].forward(h, ctx, B, ch * cw)
                skips.append((h, ch, cw))
            if ds is not None:
                h = ds.forward(h, B, ch, cw, train=False)
                ch, cw = ch // 2, cw // 2
                skips.append((h, ch, cw))
        h = self.mid[0].forward(h, None, semb, B, ch, cw)
        h = self.mid[1].forward(h, ctx, B, ch * cw)
        h = self.mid[2].forward(h, None, semb, B, ch, cw)
        for (res, att, us) in self.up:
            for j, r in enumerate(res):
                s, sh, sw = skips.pop()
                assert (sh, sw) == (ch, cw)
                h = r.forward(h, s, semb, B, ch, cw)
                if att:
                    h = att[j].forward(h, ctx, B, ch * cw)
            if us is not None:
                h = us.forward(h, B, ch, cw, train=False)
                ch, cw = ch * 2, cw * 2
        hn = self.norm_out.forward(h, None, B, ch * cw)
        self._dims = (B, H, W)
        self._daam_forward()
        return self.conv_out.forward(hn, B, ch, cw, out=self.buf("pred", B * H * W, cfg["out_channels"], dtype=F32), train=False)

    def _daam_forward(self):
        """DAAM side output (ti_cross_attn_loss.py:201-212) of every hooked cross-attention layer: S_l = Q_l K_l^T / sqrt(d) summed
        over heads is one dense [N, C] x [C, 77] product per layer, and the token-attention loss only uses the mean over layers.
        One batched GEMM per (resolution, width, batch element) into fp32 partial maps + one sum per resolution, instead of one
        32-workgroup accumulate launch per layer (60 in SDXL)."""
        rt = self.rt
        if not rt.defer_daam_scores or not rt.daam_sums:
            return
        if getattr(self, "_daam_fwd", None) is None:
            groups = {}
            for a in self.cross_attns:
                if a.hooked and getattr(a, "_qk", None) is not None:
                    groups.setdefault((a._dims[1], a.C), []).append(a)
            plan, parts = [], {}
            for (N, C), members in groups.items():
                B, _, Nk, Nkp = members[0]._dims
                part = self.buf(("daam_part", N, C), len(members), B * N, CTX_PAD, dtype=F32)
                for b in range(B):
                    items = [dict(X=a._qk[0][b * N:(b + 1) * N], W=a._qk[1][b * Nkp:(b + 1) * Nkp], C=part[i, b * N:(b + 1) * N]) for i, a in enumerate(members)]
                    plan.append((members[0].scale, items, rt.ops.GemmBatch(items, rt.device)))
                parts.setdefault(N, []).append(part)
            self._daam_fwd = (plan, parts)
        plan, parts = self._daam_fwd
        for scale, items, batch in plan:
            rt.ops.gemm(items[0]["X"], items[0]["W"], items[0]["C"], alpha=scale, batch=batch)
        for N, ps in parts.items():
            dst = rt.daam_sums[N][0]
            torch.sum(ps[0], dim=0, out=dst)
            for p_ in ps[1:]:
                dst.add_(p_.sum(0))

    # ------------------------------------------------------------------------------------ backward
    def backward(self, dpred64, dctx):
        """dpred64: d loss / d eps_hat as NHWC [B*H*W, 64] (4 real channels); dctx [B*CTX_PAD, D] is ACCUMULATED into
        (caller zeroes it).  LoRA gradients land in arena.grads."""
        rt = self.rt
        rt.gn_clear("bwd")
        rt.gn_prezero = True
        try:
            return self._backward(dpred64, dctx)
        finally:
            rt.gn_prezero = False

    def _backward(self, dpred64, dctx):
        rt, cfg = self.rt, self.cfg
        B, H, W = self._dims
        boc = cfg["block_out_channels"]
        tr = self.trainer
        if tr is not None:
            tr.zero_vector_grads()
        if rt.want_dpooled or tr is not None:
            rt.dsemb = self.buf("dsemb", B, self.tdim, zero=True)
            rt.dsemb.zero_()
            if self.temb_W is not None and "dtp_all" not in self._b:
                dtp_all = self.buf("dtp_all", B, self.temb_W.shape[0], zero=True)
                items = []
                for r in self.resnets:      # [B, cout] slices are contiguous only for B == 1 (sdlt_colsum writes a dense [B, C])
                    r._dtp_slot = dtp_all[:, r.temb_off: r.temb_off + r.cout] if B == 1 else None
                    if B == 1 and COLSUM_FUSED:
                        _, _, _, Hr, Wr = r._in
                        ns = rt.ops.groupnorm_colsum_splits(B, Hr * Wr, r.cout)
                        r._dtp_ws = torch.zeros(ns * B * r.cout, dtype=F32, device=rt.device)
                        items.append((r._dtp_ws, ns, r._dtp_slot.reshape(-1)))
                self._dtp_plan = rt.ops.ColsumFinishPlan(items, rt.device) if items else None
        dh = self.norm_out.backward(self.conv_out.backward(dpred64))
        skip_grads = []
        nlev = len(boc)
        for idx in range(nlev - 1, -1, -1):
            res, att, us = self.up[idx]
            if us is not None:
                dh = us.backward(dh)
            for j in range(len(res) - 1, -1, -1):
                if att:
                    dh = att[j].backward(dh, dctx)
                dcat = res[j].backward(dh)
                c1 = res[j].cin1
                dh = dcat[:, :c1]
                skip_grads.append(dcat[:, c1:])
        dh = self.mid[2].backward(dh)
        dh = self.mid[1].backward(dh, dctx)
        dh = self.mid[0].backward(dh)
        # skip_grads[i] belongs to the i-th pushed skip tensor (the up path pops them last-in-first-out and this
        # backward walks the up path in reverse), so the down path - also walked in reverse - pops from the end.
        pop_skip = skip_grads.pop
        for idx in range(nlev - 1, -1, -1):
            res, att, ds = self.down[idx]
            if ds is not None:
                dh = self._add(dh, pop_skip(), ("ds", idx))
                dh = ds.backward(dh)
            for j in range(len(res) - 1, -1, -1):
                dh = self._add(dh, pop_skip(), ("r", idx, j))
                if att:
                    dh = att[j].backward(dh, dctx)
                dh = res[j].backward(dh)
        # conv_in's own skip gradient and dX are not needed (its input is data, no adapter upstream) - unless conv_in itself trains
        if tr is not None:
            self.conv_in.weight_grad(self._add(dh, pop_skip(), ("cin",)))
        if rt.want_dpooled and self.temb_W is not None and B == 1:
            if getattr(self, "_dtp_plan", None) is not None:
                self._dtp_plan.run()
            rt.ops.gemm(self._b["dtp_all"], self.temb_Wt, rt.dsemb)          # dsemb = sum over the resnets of colsum(dc1) . W_temb
        if rt.want_dpooled or tr is not None:
            # emb = time_embedding(t) + add_embedding([pooled | sinusoid(time_ids)]); with LoRA only the pooled text embedding is
            # trainable upstream (textual inversion through text_encoder_2), so the timestep branch gets no backward; the
            # full fine-tune trains both MLPs.
            demb = rt.ops.map_bf16(_ops.MAP_DSILU, self._b["semb_in"], rt.dsemb, self.buf("demb", B, self.tdim))
            if cfg["addition"]:
                da1s = self.a2.backward(demb)
                da1 = rt.ops.map_bf16(_ops.MAP_DSILU, self.a1._b["y"], da1s, self.buf("da1", *da1s.shape))
                self.dadd_in = self.a1.backward(da1)
            if tr is not None:
                de1s = self.t2.backward(demb)
                self.t1.weight_grad(rt.ops.map_bf16(_ops.MAP_DSILU, self.t1._b["y"], de1s, self.buf("de1", *de1s.shape)))
        if tr is not None and not tr.defer_flush:
            tr.flush()             # every weight gradient of this pass, batched by layer shape (data parallel: the step flushes bucket by bucket)
        self._cross_kv_backward(dctx)
        self.lora_grads()

    def lora_grads(self):
        """The grouped adapter-gradient launch of this pass (+ DoRA's magnitude gradients): every dY / T / U / input buffer is complete once the
        backward pass has run; nothing else in the step reads its output before the optimizer."""
        rt = self.rt
        if self.arena is not None:
            if self._grad_plan is None:
                self._grad_plan = rt.ops.LoraGradPlan(self.arena.problems, self.arena.Rp, rt.device)
            self._grad_plan.run()
            self.arena.dora_mag_grad()

    # ------------------------------------------------------------------------------------ batched score-gradient GEMMs
    def daam_backward(self):
        """Gradient of the token-attention loss w.r.t. Q and K of every hooked cross-attention layer, BEFORE the backward
        pass: S_l = a Q_l K_l^T with one dS per resolution, so dQ_l = a dS K_l and dK_l = a dS^T Q_l are two batched GEMMs per
        (resolution, width) group written straight into the layers' dq / dk buffers; the single-pass attention backward
        then accumulates onto them (120 small launches -> 4 per batch element)."""
        rt = self.rt
        rt.daam_applied = False
        if rt.daam_grads is None or not self.cross_attns:
            return
        if getattr(self, "_daam_plan", None) is None:
            groups = {}
            for a in self.cross_attns:
                if a.hooked and a.kv_batched and a.d <= 96 and CTX_PAD <= 128 and "Qt" in a._b:
                    groups.setdefault((a._dims[1], a.C), []).append(a)
            plan = []
            for (N, C), members in groups.items():
                B, _, Nk, Nkp = members[0]._dims
                Mq, Mk = B * N, B * Nkp
                for b in range(B):
                    iq, ik = [], []
                    for a in members:
                        dq = a.buf("dq", Mq, C)
                        _, (dk, _) = a.stack.grad_slices(Mk)
                        iq.append(dict(W=a._b["Kt"][:, b * Nkp:(b + 1) * Nkp], C=dq[b * N:(b + 1) * N]))
                        ik.append(dict(W=a._b["Qt"][:, b * N:(b + 1) * N], C=dk[b * Nkp:(b + 1) * Nkp]))
                    plan.append((N, b, Nkp, members[0].scale, iq, rt.ops.GemmBatch(iq, rt.device), ik, rt.ops.GemmBatch(ik, rt.device)))
                for a in members:
                    a.daam_batched = True
            self._daam_plan = plan
        for (N, b, Nkp, scale, iq, bq, ik, bk) in self._daam_plan:
            dS, dSt = rt.daam_grads[N]
            rt.ops.gemm(dS[b * N:(b + 1) * N], iq[0]["W"], iq[0]["C"], alpha=scale, batch=bq)
            rt.ops.gemm(dSt[b * Nkp:(b + 1) * Nkp], ik[0]["W"], ik[0]["C"], alpha=scale, batch=bk)
        rt.daam_applied = bool(self._daam_plan)

    # ------------------------------------------------------------------------------------ batched cross-attention K/V
    def _cross_kv_forward(self, ctx, B):
        if not self.cross_attns or not self.cross_attns[0].kv_batched:
            return
        rt = self.rt
        if self._kv_groups is None:
            groups = {}
            for a in self.cross_attns:
                groups.setdefault((a.C, a.hooked), []).append(a)
            self._kv_groups = []
            Mk = B * CTX_PAD
            for (C, hooked), members in groups.items():
                items = []
                for a in members:
                    st = a.stack
                    y, T, _ = st.prepare(ctx)
                    it = dict(W=st.W, Adown=st.A_cat, Bup=st.B_cat, T_out=T, C=y)
                    if st.dora:
                        it["col_scale"] = st.col_scale()
                    if hooked:      # transposed K for the score side output's backward (see Attention.backward)
                        KVt = a.buf("KVt", 2 * C, _pad_to(Mk, 8))
                        a._b["Kt"] = KVt[:C]
                        it["Ct"] = KVt
                    items.append(it)
                self._kv_groups.append((members, items, rt.ops.GemmBatch(items, rt.device)))
        for members, items, batch in self._kv_groups:
            st, it = members[0].stack, items[0]
            rt.ops.gemm(ctx, st.W, it["C"], lora=(st.A_cat, st.B_cat, st.arena.scale, it["T_out"]), Ct=it.get("Ct"),
                        lora_group_n=st.N, batch=batch, **({"col_scale": it["col_scale"]} if st.dora else {}))

    def _cross_kv_backward(self, dctx):
        """dctx += sum over all cross-attention layers of [dk | dv] . [Wk ; Wv] (+ their adapters): one batched K-grouped GEMM
        per group into fp32 partials, summed once (instead of 70 read-modify-write passes over dctx in bf16)."""
        if not self.cross_attns or not self.cross_attns[0].kv_batched:
            return
        rt = self.rt
        if getattr(self, "_kv_bwd", None) is None:
            n = len(self.cross_attns)
            Mk, D = dctx.shape
            part = self.buf("dctx_parts", n, Mk, D, dtype=F32)
            self._kv_bwd, i = [], 0
            for members, _, _ in self._kv_groups:
                items = []
                for a in members:
                    st = a.stack
                    dkv, _ = st.grad_slices(Mk)
                    U = st.backward_operands(dkv)
                    items.append(dict(X=dkv, W=st.Wt_d if st.dora else st.Wt, Adown=st.Bt_cat, Bup=st.At_cat, T_out=U, C=part[i]))
                    i += 1
                self._kv_bwd.append((members, items, rt.ops.GemmBatch(items, rt.device)))
        if getattr(self, "_sum_plan", None) is None:        # partial dK / dV slabs of every layer -> bf16 dk | dv, one launch
            items = [a._sum_item for a in self.cross_attns if getattr(a, "_sum_item", None) is not None]
            self._sum_plan = rt.ops.SplitsumPlan(items, rt.device) if items else False
        if self._sum_plan:
            self._sum_plan.run()
        for members, items, batch in self._kv_bwd:
            st, it = members[0].stack, items[0]
            rt.ops.gemm(it["X"], it["W"], it["C"], lora=(it["Adown"], it["Bup"], st.arena.scale, it["T_out"]), lora_group_k=st.N, batch=batch)
        dctx.add_(self._b["dctx_parts"].sum(0).to(dctx.dtype))

    def _add(self, a, b, key):
        return self.rt.ops.add2d(a, b, self.buf(("add",) + key, *a.shape))
