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


def test_latent_cache_from_folder_and_render(tmp_path):
    """dataset.py:31-90 end to end (images on disk -> VAE encoder plan -> posterior cache) against the oracle encoder, and the
    render loop of inference.py:363-385 (sampler -> decode -> JPEG) writing the reference's file names."""
    import csv
    import numpy as np
    from PIL import Image
    from oracle import unet_ref as U
    from sd_lora_trainer_amd import sampler, topology
    from sd_lora_trainer_amd.dataset import LatentCache, prepare_image
    cfg = V.CONFIGS["tiny"]
    sd = V.init_state(cfg, seed=0)
    rng = np.random.RandomState(0)
    rows = []
    for i in range(2):
        Image.fromarray(rng.randint(0, 256, (40, 56, 3)).astype(np.uint8)).save(tmp_path / f"{i}.png")
        Image.fromarray(rng.randint(0, 256, (40, 56)).astype(np.uint8)).save(tmp_path / f"{i}_m.png")
        rows.append(dict(image_path=f"{i}.png", mask_path=f"{i}_m.png", caption="A TOK thing" if i == 0 else ""))
    with open(tmp_path / "captions.csv", "w", newline="") as fh:
        wr = csv.DictWriter(fh, fieldnames=list(rows[0]))
        wr.writeheader()
        wr.writerows(rows)
    rt = unet_mod.Runtime("cpu", 1, act_dtype=torch.float32, ops=emu_ops)
    enc = vae.VaeEncoder(rt, sd)
    cache = LatentCache.from_folder(str(tmp_path), enc, size=[32, 32], scaling_factor=0.18215, substitute_caption_map={"TOK": "<s0><s1>"})
    assert cache.captions == ["a <s0><s1> thing", ""] and len(cache) == 2
    ref = V.encode_moments(cfg, sd, prepare_image(Image.open(tmp_path / "0.png"), 32, 32))
    torch.testing.assert_close(cache.dists[0].parameters, ref, rtol=1e-3, atol=1e-4 * float(ref.abs().max()) + 1e-5)
    cap, lat, m = cache[0]
    assert lat.shape == (4, 8, 8) and m.shape == (4, 8, 8) and float(m.max()) <= 1.0

    ucfg = U.CONFIGS["tiny15"]
    rt2 = unet_mod.Runtime("cpu", 2, act_dtype=torch.float32, ops=emu_ops)
    unet = unet_mod.UNet(rt2, topology.CONFIGS["tiny15"], U.init_unet_state(ucfg, seed=0), lora_rank=4)
    unet.arena.load(U.init_lora(ucfg, 4, seed=1, b_std=0.05))
    smp = sampler.LatentSampler(rt2, unet)
    dec = vae.VaeDecoder(unet_mod.Runtime("cpu", 1, act_dtype=torch.float32, ops=emu_ops), sd)
    g = torch.Generator().manual_seed(2)
    embeds = [(torch.randn(1, 77, ucfg["cross_dim"], generator=g), torch.randn(1, 77, ucfg["cross_dim"], generator=g), None, None) for _ in range(2)]
    paths = sampler.render_images(smp, dec, embeds, (32, 32), str(tmp_path / "out"), 40, seed=7, scaling_factor=0.18215, lora_scale=0.85, n_steps=2)
    assert [p.split("/")[-1] for p in paths] == ["img_0040_0.jpg", "img_0040_1.jpg"]
    im = Image.open(paths[0])
    # (the toy VAE has 3 levels = x4, the real one x8: a 4x4 latent decodes to 16 px here)
    assert im.size == (16, 16) and im.format == "JPEG" and unet.arena.scale == 1.0
