"""Known-answer checks that pin the oracle UNet topology.  CPU only."""
import math

import torch

from oracle import unet_ref as U


def _count(cfg):
    return sum(math.prod(s) for s in U.param_shapes(cfg).values())


def test_parameter_totals():
    assert _count(U.CONFIGS["sd15"]) == 859_520_964
    assert _count(U.CONFIGS["sdxl"]) == 2_567_463_684


def test_lora_census():
    for ver, nlin, nconv, nparam in (("sd15", 128, 22, 6_414_336), ("sdxl", 560, 17, 25_425_920)):
        cfg = U.CONFIGS[ver]
        shapes = U.param_shapes(cfg)
        t = U.lora_targets(cfg)
        lin = [m for m in t if len(shapes[m + ".weight"]) == 2]
        assert (len(lin), len(t) - len(lin)) == (nlin, nconv)
        r = 16
        n = sum(r * math.prod(shapes[m + ".weight"][1:]) + shapes[m + ".weight"][0] * r for m in t)
        assert n == nparam


def test_hooked_layer_counts_and_forward_tiny():
    for ver, nhook in (("tiny15", 2 + 4), ("tinyxl", 3 + 6)):
        cfg = U.CONFIGS[ver]
        sd = U.init_unet_state(cfg, seed=0)
        lora = U.init_lora(cfg, 4, seed=1, b_std=0.05)
        B, h = 2, 16
        g = torch.Generator().manual_seed(0)
        x = torch.randn(B, 4, h, h, generator=g)
        t = torch.tensor([10, 900])
        ctx = torch.randn(B, 77, cfg["cross_dim"], generator=g)
        add = None
        if cfg["addition"]:
            add = {"text_embeds": torch.randn(B, cfg["proj_class_in"] - 6 * cfg["addition_time_embed_dim"], generator=g),
                   "time_ids": torch.tensor([[1024., 1024, 0, 0, 128, 128]] * B)}
        out, daam = U.unet_forward(cfg, sd, x, t, ctx, add, lora=lora, return_daam=True)
        assert out.shape == x.shape and torch.isfinite(out).all()
        assert len(daam) == nhook
        out0 = U.unet_forward(cfg, sd, x, t, ctx, add, lora=None)
        assert (out - out0).abs().max() > 1e-6   # adapters are live
