"""VAE (SURVEY 8f-2 decode / 8f-3 encode): known-answer parameter count of the oracle restatement, and the forward plans
(through the CPU op emulation, fp32) against it."""
import math

import pytest
import torch

from oracle import vae_ref as V
from sd_lora_trainer_amd import unet as unet_mod
from sd_lora_trainer_amd import vae

from . import emu_ops


def test_known_parameter_count():
    assert sum(math.prod(s) for s in V.param_shapes(V.CONFIGS["sd"]).values()) == 83_653_863      # AutoencoderKL of SD1.5 / SDXL


@pytest.mark.parametrize("B,h,w", [(1, 8, 8), (2, 8, 16)])
def test_vae_plans_match_oracle(B, h, w):
    cfg = V.CONFIGS["tiny"]
    sd = V.init_state(cfg, seed=0)
    g = torch.Generator().manual_seed(1)
    z = torch.randn(B, 4, h, w, generator=g)
    rt = unet_mod.Runtime("cpu", B, act_dtype=torch.float32, ops=emu_ops)
    dec, enc = vae.VaeDecoder(rt, sd), vae.VaeEncoder(rt, sd)
    img_o = V.decode(cfg, sd, z)
    img = dec.decode(z)
    f = 2 ** (len(cfg["block_out_channels"]) - 1)
    assert img.shape == (B, 3, f * h, f * w)
    torch.testing.assert_close(img, img_o, rtol=1e-3, atol=1e-4 * float(img_o.abs().max()) + 1e-5)
    x = torch.tanh(torch.randn(B, 3, f * h, f * w, generator=g))
    mom_o = V.encode_moments(cfg, sd, x)
    mom = enc.encode_moments(x)
    assert mom.shape == (B, 8, h, w)
    torch.testing.assert_close(mom, mom_o, rtol=1e-3, atol=1e-4 * float(mom_o.abs().max()) + 1e-5)
    torch.testing.assert_close(vae.postprocess(img), V.postprocess(img_o), rtol=1e-3, atol=1e-4)
