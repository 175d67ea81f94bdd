"""Validation sampling in latent space (/root/reference trainer/inference.py:180-214, 289-385): the UNet of
the training step in inference mode under the reference's render settings - Euler-discrete scheduler with trailing
timestep spacing, 25 steps, classifier-free guidance 8, LoRA adapters scaled by `sample_imgs_lora_scale` (0.75 SDXL / 0.85
SD1.5, main.py:57-61; `set_adapter_scales`, checkpoint.py:31-55) and the trigger-token strength blended between the prompt
with and without the concept (`blend_conditions`, inference.py:180-228).  Produces the final LATENTS; decoding them is the
VAE's job (vae.py).

`EulerDiscrete` restates diffusers' EulerDiscreteScheduler (0.29.2, third party, parity unpinned) for the configuration the
reference builds (`from_config(training scheduler config, timestep_spacing="trailing")`: scaled-linear betas, epsilon or
v prediction, no Karras sigmas): sigma_t = sqrt((1 - abar_t) / abar_t), timesteps = round(arange(T, 0, -T/n)) - 1, sigmas
interpolated at those timesteps plus a final 0, initial noise scaled by sigma_max (trailing spacing), model input x / sqrt(sigma^2 + 1),
x_next = x + d * (sigma_next - sigma) with d = eps (epsilon prediction).
"""
import numpy as np
import torch

from .step import ddpm_alphas_cumprod
from .unet import CTX_PAD, F32


def blend_conditions(embeds1, embeds2, lora_scale, token_scale_power=0.4, min_token_scale=0.5, token_scale=None):
    """inference.py:180-228: linear interpolation between the conditioning WITHOUT the concept (embeds1) and WITH it
    (embeds2); token_scale = min + (1 - min) * lora_scale^power unless given.  embeds = (c, uc[, pc, puc])."""
    if token_scale is None:
        token_scale = min_token_scale + (1 - min_token_scale) * lora_scale ** token_scale_power
    e1, e2 = (tuple(embeds1) + (None, None))[:4], (tuple(embeds2) + (None, None))[:4]      # always (c, uc, pc, puc) like the reference
    out = tuple(None if (a is None or b is None) else (1 - token_scale) * a + token_scale * b for a, b in zip(e1, e2))
    return out, token_scale


class EulerDiscrete:
    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, prediction_type="epsilon"):
        acp = ddpm_alphas_cumprod(num_train_timesteps, beta_start, beta_end).double().numpy()
        self.sigmas_all = np.sqrt((1 - acp) / acp)
        self.T, self.prediction_type = num_train_timesteps, prediction_type

    def set_timesteps(self, n):
        ts = np.round(np.arange(self.T, 0, -self.T / n)) - 1                       # "trailing"
        sig = np.interp(ts, np.arange(self.T), self.sigmas_all)
        self.timesteps = ts.astype(np.float32)
        self.sigmas = np.concatenate([sig, [0.0]]).astype(np.float32)
        # diffusers 0.29.2 EulerDiscreteScheduler.init_noise_sigma: max(sigmas) for timestep_spacing in ("linspace", "trailing") -
        # the reference's configuration (inference.py:358-360); sqrt(sigma_max^2 + 1) only for "leading"
        self.init_noise_sigma = float(self.sigmas.max())
        return self

    def scale_model_input(self, x, i):
        return x / float(np.sqrt(self.sigmas[i] ** 2 + 1))

    def step(self, model_out, i, x):
        s, sn = float(self.sigmas[i]), float(self.sigmas[i + 1])
        if self.prediction_type == "epsilon":
            d = model_out
        else:                                                                     # v_prediction
            x0 = model_out * (-s / (s * s + 1) ** 0.5) + x / (s * s + 1)
            d = (x - x0) / s
        return x + d * (sn - s)


class LatentSampler:
    """`pipe(prompt_embeds=c, negative_prompt_embeds=uc, ..., num_inference_steps, guidance_scale, generator)` of the
    reference's render loop, up to the latents.  `unet` is an inference instance built for batch 2 (negative | positive, the
    order diffusers concatenates them in); its LoRA arena holds the trained adapters."""

    def __init__(self, rt, unet, prediction_type="epsilon"):
        assert rt.B == 2, "classifier-free guidance runs the negative and the positive prompt as one batch of 2"
        self.rt, self.unet = rt, unet
        self.sched = EulerDiscrete(prediction_type=prediction_type)
        cfg = unet.cfg
        self.ctx = rt.zeros(2 * CTX_PAD, cfg["cross_dim"])
        self.pooled = rt.zeros(2, cfg["proj_class_in"] - 6 * cfg["addition_time_embed_dim"]) if cfg["addition"] else None

    def set_lora_scale(self, lora_scale, train_scale=None):
        """set_adapter_scales (checkpoint.py:31-55): every adapter's contribution is multiplied by lora_scale."""
        a = self.unet.arena
        if a is not None:
            if not hasattr(a, "_train_scale"):
                a._train_scale = a.scale if train_scale is None else train_scale
            a.set_scale(a._train_scale * lora_scale)      # (DoRA: the column factors follow the scale in effect)

    @torch.no_grad()
    def sample(self, embeds, h, w, *, steps=25, guidance_scale=8.0, generator=None, size=None, latents=None):
        """embeds = (c [1,77,D], uc [1,77,D], pc [1,P] | None, puc | None); h, w latent size.  Returns latents [1,4,h,w] fp32
        (still multiplied by the VAE scaling factor, as the pipeline holds them before `vae.decode(latents / scaling_factor)`)."""
        rt, u, cfg = self.rt, self.unet, self.unet.cfg
        dev = rt.device
        c, uc, pc, puc = (tuple(embeds) + (None, None))[:4]
        cv = self.ctx.view(2, CTX_PAD, -1)
        cv[0, :77].copy_(uc[0])
        cv[1, :77].copy_(c[0])
        tid = None
        if cfg["addition"]:
            self.pooled[0].copy_(puc[0])
            self.pooled[1].copy_(pc[0])
            H, W = size if size is not None else (8 * h, 8 * w)
            tid = torch.tensor([float(H), float(W), 0.0, 0.0, float(H), float(W)] * 2, device=dev)   # original_size, crop, target_size
        s = self.sched.set_timesteps(steps)
        x = latents if latents is not None else torch.randn(1, 4, h, w, generator=generator, device=dev, dtype=F32)
        x = x.to(dev, F32) * s.init_noise_sigma
        x64 = rt.zeros(2 * h * w, 64)
        for i, t in enumerate(s.timesteps):
            xin = s.scale_model_input(x, i)
            x64[:, :4] = xin.permute(0, 2, 3, 1).reshape(h * w, 4).repeat(2, 1).to(x64.dtype)
            tf = torch.full((2,), float(t), device=dev, dtype=F32)
            eps = u.forward(x64, tf, self.ctx, self.pooled, tid, B=2, H=h, W=w).view(2, h, w, 4).permute(0, 3, 1, 2)
            e = eps[0:1] + guidance_scale * (eps[1:2] - eps[0:1])
            x = s.step(e, i, x)
        return x


def render_images(sampler, decoder, embeds_list, render_size, out_dir, train_step, seed, *, scaling_factor, lora_scale,
                  n_steps=25, guidance_scale=8.0):
    """The loop of `render_images` (inference.py:363-385) after prompt encoding: one image per conditioning 4-tuple, ONE
    generator seeded once for all of them, latents -> vae.decode(latents / scaling_factor) -> [0, 1] -> JPEG quality 95 as
    `img_{train_step:04d}_{i}.jpg`; the adapters are set back to scale 1 afterwards.  render_size = (width, height)."""
    import os
    from PIL import Image
    from . import vae as _vae
    os.makedirs(out_dir, exist_ok=True)
    dev = sampler.rt.device
    gen = torch.Generator(device=dev).manual_seed(seed)
    w, h = render_size[0] // 8, render_size[1] // 8
    sampler.set_lora_scale(lora_scale)
    paths = []
    try:
        for i, embeds in enumerate(embeds_list):
            lat = sampler.sample(embeds, h, w, steps=n_steps, guidance_scale=guidance_scale, generator=gen, size=(render_size[1], render_size[0]))
            img = _vae.postprocess(decoder.decode(lat / scaling_factor))[0].permute(1, 2, 0)
            arr = (img.float().cpu().numpy() * 255).round().astype("uint8")
            paths.append(os.path.join(out_dir, f"img_{train_step:04d}_{i}.jpg"))
            Image.fromarray(arr).save(paths[-1], format="JPEG", quality=95)
    finally:
        sampler.set_lora_scale(1.0)
    return paths
