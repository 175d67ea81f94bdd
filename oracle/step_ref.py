"""ORACLE (test infrastructure only - never imported by the product path).

CPU fp32 restatement of ONE iteration of the reference's training loop, /root/reference main.py:263-382, composed of
the pieces this directory already restates (unet_ref.py, loss_ref.py) and - for the text encoders - the installed
Hugging Face `transformers` CLIP with autograd (the reference reaches the same classes through diffusers'
`encode_prompt`, trainer/inference.py:131-177).  It is what the multi-step "loss trajectory" parity tests compare the
HIP path with: same weights, same injected latents / noise / timesteps / captions, `torch.optim.AdamW` exactly as
trainer/optimizer.py:17-18,113-150 builds it.

Order of one step (main.py line numbers):
  306-308  text conditioning with autograd down to the token-embedding tables
  326      noisy = add_noise(latent, noise, t)
  329-336  model_pred = unet(...)  (+ DAAM score side outputs of the hooked attn2 layers)
  339      img loss (loss.py:127-170)
  342-345  + token_attention_loss_w * token-attention loss (only with textual inversion)
  353-356  + l1_penalty * mean |LoRA params|
  358-359  + token regulariser (std term; only while the TI learning rate is > 0)
  362-363  (loss / grad_accum).backward()
  368-371  grads of the non-trained rows of every embedding table *= 0
  381-382  optimizers: textual inversion, (text-encoder LoRA), unet; zero_grad

PARITY UNPINNED at the diffusers boundary (see unet_ref.py); the loss / regulariser / AdamW pieces are pinned by
tests/golden (see loss_ref.py).
"""
import torch

from . import loss_ref as L
from . import unet_ref as U


class RefTrainer:
    def __init__(self, cfg, sd, lora, *, text_models=None, n_tokens=3, train_ids=None, snr_gamma=5.0, l1_penalty=0.03,
                 weight_decay=0.004, ti_weight_decay=0.0, token_attention_loss_w=3e-7, ti_std_loss_w=0.01, lora_scale=1.0):
        """cfg/sd: unet_ref config + fp32 state dict; lora: module -> (A, B) (peft layout); text_models: list of HF
        CLIPTextModel[WithProjection] whose token tables already hold the n_tokens new rows LAST (embedding_handler.py:157-223)."""
        self.cfg, self.sd = cfg, sd
        # (A, B) per adapted module, or (A, B, magnitude) for DoRA (unet_ref._Ctx.linear / conv)
        self.lora = {k: tuple(t.clone().requires_grad_(True) for t in v) for k, v in lora.items()}
        self.lora_params = [t for ab in self.lora.values() for t in ab]
        self.text = text_models
        self.n_tokens, self.train_ids = n_tokens, train_ids
        self.snr_gamma, self.l1, self.ta_w, self.std_w, self.lora_scale = snr_gamma, l1_penalty, token_attention_loss_w, ti_std_loss_w, lora_scale
        self.acp = L.ddpm_alphas_cumprod().to(next(iter(sd.values())).device)
        # optimizer.py:17-18: AdamW(params, lr=1e-4 (overwritten every step), weight_decay=lora_weight_decay)
        self.opt_unet = torch.optim.AdamW(self.lora_params, lr=1e-4, weight_decay=weight_decay)
        self.opt_ti = None
        if text_models is not None:
            # optimizer.py:113-150: every parameter whose name contains "token_embedding" -> the WHOLE tables
            self.tables = [m.get_input_embeddings().weight for m in text_models]
            for m in text_models:
                for p in m.parameters():
                    p.requires_grad_(False)
            for t in self.tables:
                t.requires_grad_(True)
            self.stats = [L.DistributionStats(t.detach()[:-n_tokens].clone()) for t in self.tables]
            self.opt_ti = torch.optim.AdamW(self.tables, lr=1e-3, weight_decay=ti_weight_decay)

    def conditioning(self, ids, time_ids):
        """encode_prompt as trainer/inference.py:131-177 wires it: SD1.5 -> last_hidden_state; SDXL -> concat of both
        encoders' hidden_states[-2] + text_embeds of text_encoder_2."""
        outs = [m(input_ids=ids, output_hidden_states=True) for m in self.text]
        if self.cfg["addition"]:
            ctx = torch.cat([outs[0].hidden_states[-2], outs[1].hidden_states[-2]], dim=-1)
            return ctx, {"text_embeds": outs[1].text_embeds, "time_ids": time_ids}
        return outs[0].last_hidden_state, None

    def step(self, latent, noise, timesteps, mask, *, lr, lr_ti=0.0, ids=None, caption_token_lists=None, time_ids=None, ctx=None,
             added_cond=None, img_ratio=1.0):
        """One iteration.  With text models: ids int64 [B,77] + caption_token_lists; without: ctx / added_cond injected.
        Returns dict(img_loss, token_attention_loss, l1, reg, tot_loss, pred)."""
        out = {}
        if self.text is not None:
            ctx, added_cond = self.conditioning(ids, time_ids)
        noisy = L.add_noise(self.acp, latent, noise, timesteps)
        pred, daam = U.unet_forward(self.cfg, self.sd, noisy, timesteps, ctx, added_cond, lora=self.lora, lora_scale=self.lora_scale, return_daam=True)
        loss = L.diffusion_loss(pred, noise, noisy, mask, self.acp, timesteps, snr_gamma=self.snr_gamma)
        out["img_loss"] = float(loss.detach())
        if self.text is not None:
            ta = L.token_attention_loss(L.daam_stack([s for _, s in daam], img_ratio), mask, caption_token_lists, self.train_ids)
            out["token_attention_loss"] = float(ta.detach())
            loss = loss + self.ta_w * ta
        if self.l1 > 0.0:
            l1 = sum(p.abs().sum() for p in self.lora_params) / sum(p.numel() for p in self.lora_params)
            out["l1"] = float(l1.detach())
            loss = loss + self.l1 * l1
        if self.opt_ti is not None and lr_ti > 0.0:
            reg = torch.stack([st.std_loss(t[-self.n_tokens:]) for st, t in zip(self.stats, self.tables)]).mean()
            out["reg"] = float(self.std_w * reg.detach())
            loss = loss + self.std_w * reg
        out["tot_loss"] = float(loss.detach())
        loss.backward()
        if self.opt_ti is not None:
            out["row_grads"] = [t.grad[-self.n_tokens:].clone() for t in self.tables]
            for t in self.tables:
                t.grad.data[:-self.n_tokens, :] *= 0.0
            self.opt_ti.param_groups[0]["lr"] = lr_ti
            self.opt_ti.step()
            self.opt_ti.zero_grad()
        self.opt_unet.param_groups[0]["lr"] = lr
        out["pred"] = pred.detach()
        out["lora_grads"] = torch.cat([p.grad.reshape(-1) for p in self.lora_params]).clone()
        self.opt_unet.step()
        self.opt_unet.zero_grad()
        return out

    def gradients(self, latent, noise, timesteps, mask, *, lr_ti=1.0, ids=None, caption_token_lists=None, time_ids=None, ctx=None, added_cond=None,
                  img_ratio=1.0, bf16_faithful=False):
        """The loss and its gradients of `step` WITHOUT touching any state (no optimizer step, no .grad): dict(pred, img_loss,
        token_attention_loss, lora_grads (flat, parameter order), row_grads).  bf16_faithful: the UNet in unet_ref's bf16-faithful mode
        (every tensor the HIP path stores in bf16 is rounded where it is stored, gradients included) - the second, tighter yardstick of
        the real-topology parity tests."""
        out = {}
        if self.text is not None:
            ctx, added_cond = self.conditioning(ids, time_ids)
        noisy = L.add_noise(self.acp, latent, noise, timesteps)
        pred, daam = U.unet_forward(self.cfg, self.sd, noisy, timesteps, ctx, added_cond, lora=self.lora, lora_scale=self.lora_scale, return_daam=True,
                                    bf16_faithful=bf16_faithful)
        loss = L.diffusion_loss(pred, noise, noisy, mask, self.acp, timesteps, snr_gamma=self.snr_gamma)
        out["img_loss"] = float(loss.detach())
        if self.text is not None:
            ta = L.token_attention_loss(L.daam_stack([s for _, s in daam], img_ratio), mask, caption_token_lists, self.train_ids)
            out["token_attention_loss"] = float(ta.detach())
            loss = loss + self.ta_w * ta
        if self.l1 > 0.0:
            loss = loss + self.l1 * sum(p.abs().sum() for p in self.lora_params) / sum(p.numel() for p in self.lora_params)
        wrt = list(self.lora_params)
        if self.opt_ti is not None:
            if lr_ti > 0.0:
                loss = loss + self.std_w * torch.stack([st.std_loss(t[-self.n_tokens:]) for st, t in zip(self.stats, self.tables)]).mean()
            wrt += list(self.tables)
        grads = torch.autograd.grad(loss, wrt)
        n = len(self.lora_params)
        out["pred"] = pred.detach()
        out["lora_grads"] = torch.cat([g.reshape(-1) for g in grads[:n]])
        out["row_grads"] = [g[-self.n_tokens:].clone() for g in grads[n:]]
        return out

    def lora_flat(self):
        return torch.cat([p.detach().reshape(-1) for p in self.lora_params])

    def adam_moments(self):
        """(exp_avg, exp_avg_sq, step count) of the adapter optimizer (torch.optim.AdamW's state, optimizer.py:17-18), flat in lora_flat()'s order."""
        st = [self.opt_unet.state[p] for p in self.lora_params]
        return (torch.cat([s_["exp_avg"].reshape(-1) for s_ in st]), torch.cat([s_["exp_avg_sq"].reshape(-1) for s_ in st]), int(st[0]["step"]))
