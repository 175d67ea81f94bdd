"""Token-attention (DAAM) loss of the reference's textual-inversion path and its gradient w.r.t. the hooked
cross-attention score maps.

reference: trainer/ti_cross_attn_loss.py:239-268 (`process_and_stack_attention_scores`) + trainer/loss.py:10-80
(`compute_token_attention_loss`), weighted by config.token_attention_loss_w = 3e-7 (main.py:342-345).

Two facts make this cheap here:
  * every term of the loss depends on the stacked maps only through their MEAN over layers, and the bicubic resize is
    linear, so the hooked layers of one resolution are accumulated into one fp32 sum by the score GEMM's epilogue
    (unet.Attention) and the gradient is ONE tensor per resolution, shared by all its layers;
  * the per-caption token bookkeeping (python loops over `tokenizer.encode(caption)` in the reference) becomes three
    small device tensors, so the whole thing is hipGraph-capturable.
The remaining arithmetic is on [B, 32, 32, 77]-sized tensors (a few hundred KB) and is expressed with torch ops on the
device - plumbing-sized work; the FLOPs of this loss (the score GEMMs and their backward) run in sdlt_gemm_bf16.
"""
import math
import os

import torch
import torch.nn.functional as F

from .unet import CTX_PAD

T_TOKENS = 77
FUSED = os.environ.get("SDLT_TA_FUSED", "1") != "0"      # 0: the torch-op form below on the GPU too (A/B, and the oracle of the kernel's test)


def _gather_rows(tab, sel, dst):
    """dst[...] = tab[sel] as ONE launch (an indexing temporary + copy_ are two) when the buffer has the table's row shape and dtype."""
    if dst.dtype == tab.dtype and dst.is_contiguous() and tuple(dst.shape) == (sel.numel(), *tab.shape[1:]):
        torch.index_select(tab, 0, sel, out=dst)
    else:
        dst.copy_(tab[sel].reshape(dst.shape))


class TokenAttentionLoss:
    def __init__(self, rt, n_tok):
        self.rt, self.n_tok = rt, n_tok
        B, dev = rt.B, rt.device
        self.tok_w = torch.zeros(B, T_TOKENS, device=dev)
        self.tok_cnt = torch.ones(B, device=dev)
        self.ti_onehot = torch.zeros(B, n_tok, T_TOKENS, device=dev)
        self.has_ti = torch.zeros(B, device=dev)
        self.loss = torch.zeros(1, device=dev)
        self._bufs = {}

    def caption_table(self, token_id_lists, train_ids):
        """The per-caption constants of the loss for a whole set of captions (the job's dataset + the caption-dropout caption), on
        the device: the step then picks its batch's rows with one device-side gather (`set_from_table`) instead of rebuilding them on
        the host and copying them up every step."""
        n = len(token_id_lists)
        tabs = self._caption_rows(token_id_lists, train_ids, n)
        return tuple(t.to(self.rt.device) for t in tabs)

    def set_from_table(self, table, sel):
        """sel: int64 [B] (device) rows of `table`."""
        for dst, tab in zip((self.tok_w, self.tok_cnt, self.ti_onehot, self.has_ti), table):
            _gather_rows(tab, sel, dst)

    def set_captions(self, token_id_lists, train_ids):
        """token_id_lists[b] = tokenizer.encode(caption_b) (BOS ... EOS, unpadded) as the reference calls it
        (loss.py:32); train_ids = the textual-inversion token ids."""
        tok_w, cnt, onehot, has = self._caption_rows(token_id_lists, train_ids, self.rt.B)
        self.tok_w.copy_(tok_w)
        self.tok_cnt.copy_(cnt)
        self.ti_onehot.copy_(onehot)
        self.has_ti.copy_(has)

    def _caption_rows(self, token_id_lists, train_ids, B):
        tok_w = torch.zeros(B, T_TOKENS)
        cnt = torch.ones(B)
        onehot = torch.zeros(B, self.n_tok, T_TOKENS)
        has = torch.zeros(B)
        for b, ids in enumerate(token_id_lists):
            ids = list(ids)
            n = len(ids)
            tok_w[b, 1:n - 1] = 1.0                       # attention_maps[..., 1:len(token_indices)-1]
            cnt[b] = max(n - 2, 1)
            try:
                pos = [ids.index(t) for t in train_ids]   # loss.py:40-43: skipped unless every TI token is present
            except ValueError:
                continue
            has[b] = 1.0
            for j, p in enumerate(pos):
                onehot[b, j, p] = 1.0
        return tok_w, cnt, onehot, has

    def _bicubic_ops(self, h, w, h_out, w_out, device):
        key = (h, w, h_out, w_out)
        ops = getattr(self, "_bic", {}).get(key)
        if ops is None:
            eye_h = torch.eye(h, device=device)[None, None]
            eye_w = torch.eye(w, device=device)[None, None]
            Wh = F.interpolate(eye_h, size=(h_out, h), mode="bicubic")[0, 0].contiguous()        # [h_out, h]
            Ww = F.interpolate(eye_w, size=(w, w_out), mode="bicubic")[0, 0].t().contiguous()    # [w_out, w]
            if not hasattr(self, "_bic"):
                self._bic = {}
            ops = self._bic[key] = (Wh, Ww)
        return ops

    def _fused_plan(self, mask, img_ratio):
        """The same arithmetic as three HIP launches (csrc/daam.hip, sdlt_token_attention_loss) instead of ~90 torch ones; built once per
        (set of resolutions, mask buffer).  None when the ops table has no such kernel (CPU op emulation) or the maps do not fit it."""
        rt, B = self.rt, self.rt.B
        if not hasattr(rt.ops, "TokenAttentionPlan") or not mask.is_cuda or B > 16 or self.n_tok > 8 or len(rt.daam_sums) > 4:
            return None
        key = (tuple(sorted(rt.daam_sums)), mask.data_ptr(), tuple(ent[0].data_ptr() for _, ent in sorted(rt.daam_sums.items())), img_ratio)
        if getattr(self, "_plan_key", None) != key:
            groups_, n_layers = [], 0
            srt = sorted(rt.daam_sums.items())
            n_min = srt[0][0]
            w0 = round(math.sqrt(n_min * img_ratio))
            h0 = round(w0 / img_ratio)
            for N, (ssum, nl, _) in srt:
                w = round(math.sqrt(N * img_ratio))
                h = round(w / img_ratio)
                Wh = Ww = None
                if N != n_min:
                    Wh, Ww = self._bicubic_ops(h, w, h0, w0, ssum.device)
                groups_.append((ssum, h, w, Wh, Ww))
                n_layers += nl
            if max(h * w for _, h, w, _, _ in groups_) > 128 * 128 or h0 * w0 > 64 * 64:
                self._plan, self._plan_key = None, key
                return None
            self._plan = rt.ops.TokenAttentionPlan(groups_, B, self.n_tok, n_layers, mask, self.tok_w, self.tok_cnt, self.ti_onehot, self.has_ti, self.loss,
                                                   act_dtype=rt.act)
            self._plan_key = key
        return self._plan

    def forward_backward(self, mask, img_ratio, weight):
        """mask [B,4,H,W] fp32.  Reads rt.daam_sums, writes rt.daam_grads (d (weight*loss) / d S per resolution) and
        self.loss (the un-weighted loss value, as losses['token_attention_loss'] logs it)."""
        rt, B = self.rt, self.rt.B
        if FUSED and mask.dtype == torch.float32 and mask.is_contiguous():
            plan = self._fused_plan(mask, img_ratio)
            if plan is not None:
                rt.daam_grads = plan.run(weight)
                return self.loss
        groups = sorted(rt.daam_sums.items())      # smallest map first
        leaves, maps, n_layers = [], [], 0
        n_min = groups[0][0]
        w_min = round(math.sqrt(n_min * img_ratio))
        h_min = round(w_min / img_ratio)
        for N, (ssum, nl, _) in groups:
            s = ssum.detach().requires_grad_(True)
            leaves.append(s)
            w = round(math.sqrt(N * img_ratio))
            h = round(w / img_ratio)
            m = s.view(B, h, w, CTX_PAD)[..., :T_TOKENS]
            if N != n_min:
                # F.interpolate(mode="bicubic") is linear and separable: out = Wh . m . Ww^T with the 1-D operators
                # obtained from torch itself (identity image, other axis unscaled).  Same numbers, two small matmuls
                # instead of torch's one-thread-per-output-pixel kernel (633 us per step for a 1 MB tensor).
                Wh, Ww = self._bicubic_ops(h, w, h_min, w_min, s.device)
                m = torch.einsum("ph,bhwt->bpwt", Wh, m)
                m = torch.einsum("qw,bpwt->bpqt", Ww, m)
            maps.append(m)
            n_layers += nl
        A = sum(maps) / float(n_layers)                                   # mean over the stacked layers [B,h,w,77]
        M = F.interpolate(mask[:, 0].float().unsqueeze(1), size=(h_min, w_min)).squeeze(1)     # nearest, [B,h,w]
        mean_att = A.mean(dim=(1, 2))                                      # [B,77]
        att_l2 = (self.tok_w * torch.relu(mean_att) ** 2).sum(1) / self.tok_cnt
        r0 = 5.0 * att_l2.mean()
        heat = torch.einsum("bjt,bhwt->bjhw", self.ti_onehot, A)           # [B,n_tok,h,w]
        n_ti = self.has_ti.sum()
        denom = torch.clamp(n_ti, min=1.0) * self.n_tok * h_min * w_min
        hb = self.has_ti.view(B, 1, 1, 1)
        Mb = M.unsqueeze(1)
        r1 = (hb * torch.relu(heat * Mb) ** 2).sum() / denom
        r2 = 2.0 * (hb * torch.relu(heat * (1 - Mb) + 10) ** 2).sum() / denom
        tok_means = heat.mean(dim=(2, 3))                                  # [B,n_tok]
        var = tok_means.var(dim=1) if self.n_tok > 1 else torch.zeros(B, device=A.device)
        r3 = (self.has_ti * var).sum() / torch.clamp(n_ti, min=1.0)
        gate = (n_ti > 0).float()                                          # loss.py:55-56: 0 if no caption holds the TI tokens
        loss = gate * (r0 + r1 + r2 + r3)
        grads = torch.autograd.grad(loss * weight, leaves)
        self.loss.copy_(loss.detach().reshape(1))
        out = {}
        for (N, _), g in zip(groups, grads):
            key = ("dS", N)
            if key not in self._bufs:
                self._bufs[key] = (torch.zeros(B * N, CTX_PAD, dtype=rt.act, device=rt.device),
                                   torch.zeros(B * CTX_PAD, N, dtype=rt.act, device=rt.device))
            dS, dSt = self._bufs[key]
            dS.copy_(g)
            dSt.view(B, CTX_PAD, N).copy_(g.view(B, N, CTX_PAD).transpose(1, 2))
            out[N] = (dS, dSt)
        rt.daam_grads = out
        return self.loss
