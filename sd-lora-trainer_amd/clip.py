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
        self.layers = [ClipLayer(rt, f"{pre}encoder.layers.{i}", sd, heads, act, arena, lora_prefix) for i in range(self.n_run)]
        self.final_ln = LayerNorm(rt, pre + "final_layer_norm", sd) if (mode == "last" or with_projection) else None
        self.proj = Linear(rt, "text_projection", sd) if with_projection else None
        self.train_ids = torch.arange(self.V - n_train, self.V, dtype=torch.int64, device=rt.device)

    def forward(self, ids, B, hidden_out=None, pool_rows=None, hidden_only=False):
        """ids int64 [B,77] (device).  hidden_out: optional [B*TP, D] (strided) destination of the hidden states.
        pool_rows int64 [B]: row index b*TP + pool position (HF: argmax / first EOS) for the pooled output.
        hidden_only: the caller needs no pooled output - nothing above the hidden state is run."""
        rt = self.rt
        x = rt.ops.embed_gather(self.table, ids, self.pos, self.buf("x0", B * TP, self.D), B=B, T=T_TOKENS, Tp=TP)
        self._ids, self._B = ids, B
        hidden = None
        for i, layer in enumerate(self.layers):
            is_hidden = (i + 1 == self.n_hidden) and self.mode == "penultimate"
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
        """d_hidden [B*TP, D] (strided view ok; pad rows must be zero), d_pooled [B,P] or None -> grad_rows fp32 [n_train, D]."""
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
            dx = self.layers[i].backward(dx)
        return rt.ops.embed_grad(dx, self._ids, self.train_ids, grad_rows, B=B, T=T_TOKENS, Tp=TP, accumulate=accumulate)


# OpenAI CLIP-L/14 text tower and OpenCLIP bigG/14 text tower as wired by SD1.5 / SDXL
CLIP_L = dict(heads=12, act="quick_gelu")
CLIP_G = dict(heads=20, act="gelu")
