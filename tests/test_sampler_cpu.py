"""Validation sampling: the reference's own `blend_conditions` (golden vector), known answers of the Euler
"trailing" schedule, and the latent sampler of the product (through the CPU op emulation) against the fp32 oracle loop."""
import math
import os

import pytest
import torch

from oracle import sampler_ref as SR
from oracle import unet_ref as U
from sd_lora_trainer_amd import sampler, topology
from sd_lora_trainer_amd import unet as unet_mod

from . import emu_ops


def test_blend_conditions_golden(golden_dir):
    for c in torch.load(os.path.join(golden_dir, "blend_conditions.pt"), weights_only=False):
        out, ts = sampler.blend_conditions(c["e1"], c["e2"], c["lora_scale"], token_scale=c["token_scale_in"])
        assert ts == pytest.approx(c["token_scale"], rel=1e-12)
        assert len(out) == len(c["out"])
        for a, b in zip(out, c["out"]):
            assert (a is None) == (b is None)
            if a is not None:
                torch.testing.assert_close(a, b, rtol=1e-6, atol=1e-7)
    _, ts = sampler.blend_conditions((torch.zeros(1),) * 2, (torch.ones(1),) * 2, 0.75)
    assert ts == pytest.approx(0.5 + 0.5 * 0.75 ** 0.4)            # 0.5 + 0.5 s^0.4


def test_euler_trailing_known_answers():
    s = sampler.EulerDiscrete().set_timesteps(25)
    assert s.timesteps[0] == 999 and s.timesteps[-1] == 39 and len(s.timesteps) == 25 and len(s.sigmas) == 26
    assert float(s.sigmas[0]) == pytest.approx(14.6146, abs=2e-4)    # sigma_max of the SD scaled-linear schedule
    assert s.sigmas[-1] == 0.0 and all(a > b for a, b in zip(s.sigmas, s.sigmas[1:]))
    assert s.init_noise_sigma == pytest.approx(14.6146, abs=2e-4)     # trailing spacing: max(sigmas), NOT sqrt(sigma_max^2 + 1)
    ts, sig = SR.euler_trailing(25)
    assert list(ts) == [int(t) for t in s.timesteps]
    torch.testing.assert_close(torch.tensor(sig, dtype=torch.float32), torch.tensor(s.sigmas), rtol=1e-6, atol=0)
    s30 = sampler.EulerDiscrete().set_timesteps(30)                  # 30 steps for the final render (main.py)
    assert s30.timesteps[0] == 999 and len(s30.timesteps) == 30
    # one Euler step on a known input: x + eps * (sigma_1 - sigma_0)
    x, e = torch.ones(1, 4, 2, 2), torch.full((1, 4, 2, 2), 2.0)
    torch.testing.assert_close(s.step(e, 0, x), x + 2.0 * float(s.sigmas[1] - s.sigmas[0]))


@pytest.mark.parametrize("version", ["tiny15", "tinyxl"])
def test_latent_sampler_matches_oracle(version):
    cfg, h, rank, steps, scale = U.CONFIGS[version], 8, 4, 5, 0.75
    sd = U.init_unet_state(cfg, seed=0)
    lora = U.init_lora(cfg, rank, seed=1, b_std=0.05)
    g = torch.Generator().manual_seed(5)
    D = cfg["cross_dim"]
    P = cfg["proj_class_in"] - 6 * cfg["addition_time_embed_dim"] if cfg["addition"] else 0
    mk = lambda *s: torch.randn(*s, generator=g)  # noqa: E731
    embeds = (mk(1, 77, D), mk(1, 77, D)) + ((mk(1, P), mk(1, P)) if cfg["addition"] else (None, None))
    noise = mk(1, 4, h, h)
    ref = SR.sample_latents(cfg, sd, lora, scale, embeds, noise, steps, guidance_scale=8.0)

    rt = unet_mod.Runtime("cpu", 2, act_dtype=torch.float32, ops=emu_ops)
    unet = unet_mod.UNet(rt, topology.CONFIGS[version], sd, lora_rank=rank)
    unet.arena.load(lora)
    smp = sampler.LatentSampler(rt, unet)
    smp.set_lora_scale(scale)
    got = smp.sample(embeds, h, h, steps=steps, guidance_scale=8.0, latents=noise)
    torch.testing.assert_close(got, ref, rtol=2e-3, atol=2e-3 * float(ref.abs().max()))
    smp.set_lora_scale(1.0)
    assert unet.arena.scale == 1.0
