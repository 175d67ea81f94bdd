"""ORACLE (test infrastructure only - never imported by the product path).

fp32 restatement of the reference's validation render up to the latents (trainer/inference.py:289-385 -> diffusers
StableDiffusion(XL)Pipeline.__call__ with an EulerDiscreteScheduler(timestep_spacing="trailing"), guidance_scale 8):

    latents = randn * init_noise_sigma
    for t in timesteps:  x_in = cat([latents] * 2) / sqrt(sigma_t^2 + 1)
                         eps_u, eps_c = unet(x_in, t, cat([negative, positive]))      (diffusers' batch order)
                         eps = eps_u + g (eps_c - eps_u) ;  latents += eps * (sigma_next - sigma_t)

PARITY UNPINNED at the diffusers boundary (0.29.2 is neither vendored nor installed).  Known answers that ARE pinned in
tests/test_sampler_cpu.py: sigma_max of the SD scaled-linear schedule = 14.6146, trailing timesteps of 25 steps start at 999
and end at 39, init_noise_sigma = sigma_max (diffusers returns max(sigmas) for "linspace"/"trailing" spacing and
sqrt(sigma_max^2 + 1) only for "leading"; the reference sets timestep_spacing="trailing", inference.py:358-360).
"""
import numpy as np
import torch

from . import loss_ref as L
from . import unet_ref as U


def euler_trailing(n, T=1000):
    acp = L.ddpm_alphas_cumprod(T).double().numpy()
    sig_all = ((1 - acp) / acp) ** 0.5
    timesteps = np.round(np.arange(T, 0, -T / n)).astype(np.int64) - 1
    sigmas = np.interp(timesteps, np.arange(T), sig_all)
    return timesteps, np.concatenate([sigmas, [0.0]])


def sample_latents(cfg, sd, lora, lora_scale, embeds, noise, steps, guidance_scale=8.0, size=None):
    """lora: module -> (A, B) (peft layout) or None; the adapters enter with weight lora_scale (set_adapters)."""
    c, uc, pc, puc = (tuple(embeds) + (None, None))[:4]
    timesteps, sigmas = euler_trailing(steps)
    x = noise.float() * float(sigmas.max())                 # init_noise_sigma for trailing spacing
    h, w = x.shape[-2:]
    ctx = torch.cat([uc, c], 0)
    add = None
    if cfg["addition"]:
        H, W = size if size is not None else (8 * h, 8 * w)
        add = {"text_embeds": torch.cat([puc, pc], 0), "time_ids": torch.tensor([[float(H), float(W), 0.0, 0.0, float(H), float(W)]] * 2)}
    lora_s = None if lora is None else {k: (A, B * lora_scale) for k, (A, B) in lora.items()}
    with torch.no_grad():
        for i, t in enumerate(timesteps):
            xin = torch.cat([x, x], 0) / float((sigmas[i] ** 2 + 1) ** 0.5)
            eps = U.unet_forward(cfg, sd, xin, torch.tensor([int(t)] * 2), ctx, add, lora=lora_s)
            e = eps[0:1] + guidance_scale * (eps[1:2] - eps[0:1])
            x = x + e * float(sigmas[i + 1] - sigmas[i])
    return x
