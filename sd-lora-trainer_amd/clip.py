"""CLIP text encoders (CLIP-L for SD1.5/SDXL, OpenCLIP-bigG with projection for SDXL) as an explicit forward /
dX-backward plan over the HIP kernels, down to the gradient of the TRAINABLE token-embedding rows.

Drop-in for the text-conditioning forward the reference runs *with autograd* inside every training step
(/root/reference trainer/inference.py:131-177 `get_conditioning_signals` -> diffusers `pipe.encode_prompt`, called from
main.py:306-308) and for the part of `loss.backward()` (main.py:363) that reaches `token_embedding.weight`; the
reference then zeroes every gradient row but the last n_tokens (main.py:368-371) - here only those rows are ever
computed (`sdlt_embed_grad`).

Weights use the Hugging Face `CLIPTextModel(WithProjection)` state-dict names.  Sequences are stored as TP = 128 rows per
batch (77 valid) so the hidden states can be written straight into the UNet's conditioning buffer.
"""
import math
import os

import torch

from . import ops as _ops
from .unet import F32, LayerNorm, Linear, StackedLinear, _Module

T_TOKENS = 77
TP = 128


# activation of the CLIP MLP as GEMM epilogues (fc1: act side output, fc2 dX: x act') or as element-wise launches; SDLT_CLIP_ACT_EPI=0 / 1
ACT_EPILOGUE = os.environ.get("SDLT_CLIP_ACT_EPI", "1") != "0"


class _Renamed(dict):
    """state-dict view that resolves `<lora name>.weight` to the checkpoint key: the LoRA entry of a text-encoder
    projection is named with the encoder's prefix (`text_encoder.` / `text_encoder_2.`) so that both encoders can share
    one arena, while the weights are looked up under their Hugging Face names."""

    def __init__(self, sd, prefix):
        self.sd, self.prefix = sd, prefix

    def __getitem__(self, k):
        return self.sd[k[len(self.prefix):]]

    def get(self, k, default=None):
        return self.sd.get(k[len(self.prefix):], default)


class ClipLayer(_Module):
    def __init__(self, rt, name, sd, heads, act, arena=None, lora_prefix=""):
        super().__init__(rt, name)
        self.ln1 = LayerNorm(rt, name + ".layer_norm1", sd)
        # optional text-encoder LoRA on q/k/v/out_proj (trainer/optimizer.py:157-167)
        psd = _Renamed(sd, lora_prefix)
        self.q = Linear(rt, lora_prefix + name + ".self_attn.q_proj", psd, arena)
        self.k = Linear(rt, lora_prefix + name + ".self_attn.k_proj", psd, arena)
        self.v = Linear(rt, lora_prefix + name + ".self_attn.v_proj", psd, arena)
        self.qkv = StackedLinear(rt, name + ".self_attn.qkv", [self.q, self.k, self.v])
        self.o = Linear(rt, lora_prefix + name + ".self_attn.out_proj", psd, arena)
        self.ln2 = LayerNorm(rt, name + ".layer_norm2", sd)
        self.fc1 = Linear(rt, name + ".mlp.fc1", sd)
        self.fc2 = Linear(rt, name + ".mlp.fc2", sd)
        self.heads, self.D = heads, self.q.N
        self.d = self.D // heads
        self.scale = 1.0 / math.sqrt(self.d)
        self.act, self.dact = (_ops.MAP_QGELU, _ops.MAP_DQGELU) if act == "quick_gelu" else (_ops.MAP_GELU, _ops.MAP_DGELU)
        self.act_kind = "quick_gelu" if act == "quick_gelu" else "gelu"

    def _akw(self, B):
        return dict(B=B, H=self.heads, Nq=T_TOKENS, Nk=T_TOKENS, Nqp=TP, Nkp=TP, d=self.d, scale=self.scale, causal=True)

    def forward(self, x, B, out=None):
        rt, M, D = self.rt, B * TP, self.D
        n1 = self.ln1.forward(x)
        q, k, v = self.qkv.forward(n1)
        O, L = self.buf("O", M, D), self.buf("L", B * self.heads * T_TOKENS, dtype=F32)
        rt.ops.attn_fwd(q, k, v, None, O, L, **self._akw(B))
        x1 = self.o.forward(O, residual=x)
        # the activation leaves fc1's epilogue (sdlt_gemm_params.epi_op 3), its derivative is the epilogue of fc2's dX GEMM (4)
        a = self.buf("a", M, self.fc1.N)
        if ACT_EPILOGUE:
            self.fc1.forward(self.ln2.forward(x1), act_out=(self.act_kind, a))
        else:
            rt.ops.map_bf16(self.act, self.fc1.forward(self.ln2.forward(x1)), None, a)
        self._B = B
        return self.fc2.forward(a, residual=x1, out=out)

    def backward(self, dx2):
        rt, B, D = self.rt, self._B, self.D
        M = B * TP
        if ACT_EPILOGUE:
            df = self.fc2.backward(dx2, dact_in=(self.act_kind, self.fc1._b["y"]))
        else:
            da = self.fc2.backward(dx2)
            df = rt.ops.map_bf16(self.dact, self.fc1._b["y"], da, self.buf("df", *da.shape))
        dx1 = self.ln2.backward(self.fc1.backward(df), dres=dx2)
        dO = self.o.backward(dx1)
        dqkv, (dq, dk, dv) = self.qkv.grad_slices(M)
        rt.ops.attn_bwd(self.q._b["y"], self.k._b["y"], self.v._b["y"], None, None, self._b["O"], self._b["L"], dO,
                        None, self.buf("Dd", B * self.heads * T_TOKENS, dtype=F32), dq, dk, dv, **self._akw(B))
        if self.qkv.has_lora and not self.qkv.kgrouped:     # adapter rank > 16: member-wise dX, summed through the residual input
            dn1 = self.q.backward(dq)
            dn1 = self.k.backward(dk, dres=dn1, key="dx2")
            dn1 = self.v.backward(dv, dres=dn1, key="dx3")
        else:
            dn1 = self.qkv.backward(dqkv)
        return self.ln1.backward(dn1, dres=dx1)


# Fused text-encoder layers (row-strip products, sdlt_strip_gemm): SDLT_CLIP_FUSED=0 keeps the tiled-GEMM plan above; the strip kernel
# re-streams the weights once per batch element, so beyond SDLT_CLIP_FUSED_MAXB images per step the tiled plan (M = B * 128 rows) is used
FUSED = os.environ.get("SDLT_CLIP_FUSED", "1") != "0"
FUSED_MAXB = int(os.environ.get("SDLT_CLIP_FUSED_MAXB", "4"))


def fused_ok(rt, B, D, F_, arena):
    """The fused plan covers frozen text encoders (no adapters: text-encoder LoRA keeps the fused-LoRA tiled GEMMs) whose widths the
    strip kernel takes (K % 256 == 0; the CPU emulation of the op contracts takes any)."""
    if not FUSED or arena is not None or not hasattr(rt.ops, "strip_gemm") or B > FUSED_MAXB:
        return False
    return bool(getattr(rt.ops, "STRIP_ANY_SHAPE", False)) or (D % 256 == 0 and F_ % 256 == 0)


class FusedClipLayer(_Module):
    """One CLIPEncoderLayer as 5 launches forward, 7 backward (ClipLayer: 7 / 9 of the tiled kernels, each 9-18 us on 80 rows):
      forward   [LN1 + q|k|v] -> causal attention -> [out_proj + residual] -> [LN2 + fc1 + act] -> [fc2 + residual]
      backward  [dX fc2 * act'] -> [dX fc1] -> LN2 backward (+ residual gradient) -> [dX out_proj] -> attention backward
                -> [dX q|k|v] -> LN1 backward (+ residual gradient)
    every [...] one sdlt_strip_gemm launch: the LayerNorms in front of q|k|v and fc1 are folded into the products (ops.fold_layernorm),
    no normalised activations are stored; the LayerNorm backward works from the raw rows and the (mean, rstd) the product left.
    Only the 77 valid rows of a sequence are computed or written: every buffer is zero-initialised and its pad rows stay zero."""

    def __init__(self, rt, name, sd, heads, act):
        super().__init__(rt, name)
        dev, dt = rt.device, rt.act
        g = lambda k: sd[name + k].to(dev, F32)  # noqa: E731
        wq, wk, wv = g(".self_attn.q_proj.weight"), g(".self_attn.k_proj.weight"), g(".self_attn.v_proj.weight")
        wqkv = torch.cat([wq, wk, wv], 0)
        bqkv = torch.cat([g(".self_attn.q_proj.bias"), g(".self_attn.k_proj.bias"), g(".self_attn.v_proj.bias")])
        self.g1, self.be1 = g(".layer_norm1.weight").contiguous(), g(".layer_norm1.bias").contiguous()
        self.g2, self.be2 = g(".layer_norm2.weight").contiguous(), g(".layer_norm2.bias").contiguous()
        self.eps = 1e-5
        fold = rt.ops.fold_layernorm
        self.Wqkv_g, self.qkv_c1, self.qkv_c2 = fold(wqkv, bqkv, self.g1, self.be1, dtype=dt)
        self.Wqkv_t = wqkv.t().to(dt).contiguous()
        wo = g(".self_attn.out_proj.weight")
        self.Wo, self.Wo_t, self.bo = wo.to(dt).contiguous(), wo.t().to(dt).contiguous(), g(".self_attn.out_proj.bias").contiguous()
        w1, w2 = g(".mlp.fc1.weight"), g(".mlp.fc2.weight")
        self.W1_g, self.fc1_c1, self.fc1_c2 = fold(w1, g(".mlp.fc1.bias"), self.g2, self.be2, dtype=dt)
        self.W1_t = w1.t().to(dt).contiguous()
        self.W2, self.W2_t, self.b2 = w2.to(dt).contiguous(), w2.t().to(dt).contiguous(), g(".mlp.fc2.bias").contiguous()
        self.D, self.F, self.heads = wo.shape[0], w1.shape[0], heads
        self.d = self.D // heads
        self.scale = 1.0 / math.sqrt(self.d)
        self.act_kind = "quick_gelu" if act == "quick_gelu" else "gelu"

    def _akw(self, B):
        return dict(B=B, H=self.heads, Nq=T_TOKENS, Nk=T_TOKENS, Nqp=TP, Nkp=TP, d=self.d, scale=self.scale, causal=True)

    def zbuf(self, key, *shape, dtype=None):
        return self.buf(key, *shape, dtype=dtype, zero=True)

    def forward_steps(self, x, B, out=None):
        """Generator form of forward: yields after every launch (ops.run_paired pairs the launches of two encoders' layers)."""
        ops, M, D, F_ = self.rt.ops, B * TP, self.D, self.F
        kw = dict(B=B, T=T_TOKENS, Tp=TP)
        self._x, self._B = x, B
        qkv = ops.strip_gemm(x, self.Wqkv_g, self.zbuf("qkv", M, 3 * D), ln=(self.qkv_c1, self.qkv_c2, self.eps),
                             stats=self.zbuf("st1", M * 2, dtype=F32), **kw)
        yield
        q, k, v = qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:]
        O, L = self.zbuf("O", M, D), self.zbuf("L", B * self.heads * T_TOKENS, dtype=F32)
        ops.attn_fwd(q, k, v, None, O, L, **self._akw(B))
        yield
        x1 = ops.strip_gemm(O, self.Wo, self.zbuf("x1", M, D), bias=self.bo, residual=x, **kw)
        yield
        a = self.zbuf("a", M, F_)
        ops.strip_gemm(x1, self.W1_g, self.zbuf("y1", M, F_), ln=(self.fc1_c1, self.fc1_c2, self.eps), stats=self.zbuf("st2", M * 2, dtype=F32),
                       act_out=(self.act_kind, a), **kw)
        yield
        y = ops.strip_gemm(a, self.W2, out if out is not None else self.zbuf("x2", M, D), bias=self.b2, residual=x1, **kw)
        yield
        return y

    def forward(self, x, B, out=None):
        return _drain(self.forward_steps(x, B, out))

    def backward_steps(self, dx2):
        ops, B, D, F_ = self.rt.ops, self._B, self.D, self.F
        M = B * TP
        kw = dict(B=B, T=T_TOKENS, Tp=TP)
        b = self._b
        df = ops.strip_gemm(dx2, self.W2_t, self.zbuf("df", M, F_), dact_in=(self.act_kind, b["y1"]), **kw)
        yield
        # the two long-K input gradients (K = mlp width, 3 D) are cut into K slices whose fp32 tiles the LayerNorm backward behind them adds
        # in its prologue: a split without a seam (an in-kernel last-arriver reduction costs 4 - 5 us, the launch boundary nothing)
        S2 = ops.strip_partial_splits(D, F_, B)
        dn2 = ops.strip_gemm(df, self.W1_t, None, partial=self.zbuf("dn2_p", S2, M, D, dtype=F32), **kw)
        yield
        dx1 = ops.layernorm_bwd(b["x1"], None, self.zbuf("dx1", M, D), b["st2"], gamma=self.g2, dres=dx2, dy_slabs=dn2)
        yield
        dO = ops.strip_gemm(dx1, self.Wo_t, self.zbuf("dO", M, D), **kw)
        yield
        qkv, dqkv = b["qkv"], self.zbuf("dqkv", M, 3 * D)
        ops.attn_bwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], None, None, b["O"], b["L"], dO, None,
                     self.zbuf("Dd", B * self.heads * T_TOKENS, dtype=F32), dqkv[:, :D], dqkv[:, D:2 * D], dqkv[:, 2 * D:], **self._akw(B))
        yield
        S1 = ops.strip_partial_splits(D, 3 * D, B)
        dn1 = ops.strip_gemm(dqkv, self.Wqkv_t, None, partial=self.zbuf("dn1_p", S1, M, D, dtype=F32), **kw)
        yield
        dx = ops.layernorm_bwd(self._x, None, self.zbuf("dx", M, D), b["st1"], gamma=self.g1, dres=dx1, dy_slabs=dn1)
        yield
        return dx

    def backward(self, dx2):
        return _drain(self.backward_steps(dx2))


def _drain(gen):
    while True:
        try:
            next(gen)
        except StopIteration as e:
            return e.value


def _sub(gen, key):
    """Re-yield the steps of a layer generator tagged with `key` (the layer index); evaluates to the layer's result."""
    while True:
        try:
            next(gen)
        except StopIteration as e:
            return e.value
        yield key


class ClipTextEncoder(_Module):
    """mode "last": hidden = final_layer_norm(layer_L)            (SD1.5 prompt embeds)
       mode "penultimate": hidden = output of layer L-1 (HF hidden_states[-2], no final LN)   (SDXL prompt embeds)
       with_projection: pooled = text_projection(final_layer_norm(layer_L)[pool position])     (SDXL text_encoder_2)"""

    def __init__(self, rt, name, sd, *, heads, act, mode, with_projection, n_train, arena=None, lora_prefix=""):
        """arena: optional unet.LoraArena shared by the text encoders (text-encoder LoRA, trainer/optimizer.py:157-202);
        its entries are named lora_prefix + the Hugging Face module path."""
        super().__init__(rt, name)
        # transformers 4.x (the reference's pin) prefixes every key with "text_model."; 5.x drops it for CLIPTextModel
        pre = "text_model." if "text_model.embeddings.token_embedding.weight" in sd else ""
        tab = sd[pre + "embeddings.token_embedding.weight"]
        self.V, self.D = tab.shape
        self.n_train = n_train
        self.table = tab.to(rt.device, rt.act).contiguous()           # last n_train rows are refreshed from the TI arena
        self.pos = sd[pre + "embeddings.position_embedding.weight"].to(rt.device, rt.act).contiguous()
        nl = 0
        while f"{pre}encoder.layers.{nl}.layer_norm1.weight" in sd:
            nl += 1
        self.mode, self.with_projection = mode, with_projection
        # layers whose output is never consumed are not built (SDXL CLIP-L: the last layer only feeds an unused pooled output)
        self.n_run = nl if (mode == "last" or with_projection) else nl - 1
        self.n_hidden = nl if mode == "last" else nl - 1              # hidden state = output of this many layers
        mlp = sd[f"{pre}encoder.layers.0.mlp.fc1.weight"].shape[0] if nl else 0
        self.fused = nl > 0 and fused_ok(rt, rt.B, self.D, mlp, arena)
        if self.fused:
            self.layers = [FusedClipLayer(rt, f"{pre}encoder.layers.{i}", sd, heads, act) for i in range(self.n_run)]
        else:
            self.layers = [ClipLayer(rt, f"{pre}encoder.layers.{i}", sd, heads, act, arena, lora_prefix) for i in range(self.n_run)]
        self.final_ln = LayerNorm(rt, pre + "final_layer_norm", sd) if (mode == "last" or with_projection) else None
        self.proj = Linear(rt, "text_projection", sd) if with_projection else None
        self.train_ids = torch.arange(self.V - n_train, self.V, dtype=torch.int64, device=rt.device)

    def forward(self, ids, B, hidden_out=None, pool_rows=None, hidden_only=False):
        return _drain(self.forward_steps(ids, B, hidden_out=hidden_out, pool_rows=pool_rows, hidden_only=hidden_only))

    def forward_steps(self, ids, B, hidden_out=None, pool_rows=None, hidden_only=False):
        """Generator (yields the layer index after every launch of a fused layer; ops.run_paired pairs two encoders' launches).
        ids int64 [B,77] (device).  hidden_out: optional [B*TP, D] (strided) destination of the hidden states.
        pool_rows int64 [B]: row index b*TP + pool position (HF: argmax / first EOS) for the pooled output.
        hidden_only: the caller needs no pooled output - nothing above the hidden state is run."""
        rt = self.rt
        x = rt.ops.embed_gather(self.table, ids, self.pos, self.buf("x0", B * TP, self.D), B=B, T=T_TOKENS, Tp=TP)
        self._ids, self._B = ids, B
        hidden = None
        for i, layer in enumerate(self.layers):
            is_hidden = (i + 1 == self.n_hidden) and self.mode == "penultimate"
            if self.fused:
                x = yield from _sub(layer.forward_steps(x, B, out=hidden_out if is_hidden else None), i)
            else:
                x = layer.forward(x, B, out=hidden_out if is_hidden else None)
            if is_hidden:
                hidden = x
                if hidden_only:
                    return hidden, None
        pooled = None
        if self.final_ln is not None:
            fin = self.final_ln.forward(x, out=hidden_out if self.mode == "last" else None)
            if self.mode == "last":
                hidden = fin
            if self.with_projection and not hidden_only:
                self._pool_rows = pool_rows
                pin = self.buf("pool_in", B, self.D)
                pin.copy_(fin[pool_rows])
                pooled = self.proj.forward(pin, train=False)
        return hidden, pooled

    def backward(self, d_hidden, d_pooled, grad_rows, accumulate=False):
        return _drain(self.backward_steps(d_hidden, d_pooled, grad_rows, accumulate=accumulate))

    def backward_steps(self, d_hidden, d_pooled, grad_rows, accumulate=False):
        """Generator form (see forward_steps).  d_hidden [B*TP, D] (strided view ok; pad rows must be zero), d_pooled [B,P] or None -> grad_rows fp32 [n_train, D]."""
        rt, B = self.rt, self._B
        dx = None
        if self.final_ln is not None:
            dfin = None
            if self.with_projection and d_pooled is not None:
                dpin = self.proj.backward(d_pooled)
                dfin = self.buf("dfin", B * TP, self.D, zero=True)
                dfin.zero_()
                dfin[self._pool_rows] = dpin
            if self.mode == "last":
                dfin = d_hidden if dfin is None else rt.ops.add2d(dfin, d_hidden, self.buf("dfin2", B * TP, self.D))
            if dfin is not None:
                dx = self.final_ln.backward(dfin)
        for i in range(self.n_run - 1, -1, -1):
            if self.mode == "penultimate" and i + 1 == self.n_hidden:
                dx = d_hidden if dx is None else rt.ops.add2d(dx, d_hidden, self.buf("dxh", B * TP, self.D))
            if dx is None:
                continue      # layers above the hidden state with no pooled gradient: nothing flows
            if self.fused:
                dx = yield from _sub(self.layers[i].backward_steps(dx), i)
            else:
                dx = self.layers[i].backward(dx)
        return rt.ops.embed_grad(dx, self._ids, self.train_ids, grad_rows, B=B, T=T_TOKENS, Tp=TP, accumulate=accumulate)


# OpenAI CLIP-L/14 text tower and OpenCLIP bigG/14 text tower as wired by SD1.5 / SDXL
CLIP_L = dict(heads=12, act="quick_gelu")
CLIP_G = dict(heads=20, act="gelu")
