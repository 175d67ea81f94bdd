"""ORACLE (test infrastructure only - never imported by the product path).

fp32 restatement of diffusers' AutoencoderKL (0.29.2, third party; not vendored, not installed -> PARITY UNPINNED at that
boundary) as the reference uses it: `vae.encode(image).latent_dist` once per training image (trainer/dataset.py:141-157)
and `vae.decode(latents / scaling_factor)` inside the validation render (trainer/inference.py:289-385 -> pipeline call).
State-dict names and topology follow the diffusers model: encoder = conv_in, 4 down blocks of 2 ResNets (+ a stride-2 conv
padded right/bottom only), mid (ResNet, single-head attention, ResNet), GroupNorm+SiLU, conv_out to 2*latent channels,
quant_conv 1x1; decoder = post_quant_conv 1x1, conv_in, mid, 4 up blocks of 3 ResNets (+ nearest-2x upsample + conv), GroupNorm
+SiLU, conv_out.  All GroupNorms have 32 groups and eps 1e-6.  Known-answer pin: the SD/SDXL configuration has 83,653,863
parameters (tests/test_vae_cpu.py)."""
import math
from collections import OrderedDict

import torch
import torch.nn.functional as F

CONFIGS = {
    "sd": dict(block_out_channels=(128, 256, 512, 512), layers_per_block=2, latent_channels=4, in_channels=3),
    "tiny": dict(block_out_channels=(64, 64, 128), layers_per_block=1, latent_channels=4, in_channels=3),
}


def _resnet_shapes(p, name, cin, cout):
    p[f"{name}.norm1.weight"], p[f"{name}.norm1.bias"] = (cin,), (cin,)
    p[f"{name}.conv1.weight"], p[f"{name}.conv1.bias"] = (cout, cin, 3, 3), (cout,)
    p[f"{name}.norm2.weight"], p[f"{name}.norm2.bias"] = (cout,), (cout,)
    p[f"{name}.conv2.weight"], p[f"{name}.conv2.bias"] = (cout, cout, 3, 3), (cout,)
    if cin != cout:
        p[f"{name}.conv_shortcut.weight"], p[f"{name}.conv_shortcut.bias"] = (cout, cin, 1, 1), (cout,)


def _mid_shapes(p, name, c):
    _resnet_shapes(p, f"{name}.resnets.0", c, c)
    a = f"{name}.attentions.0"
    p[f"{a}.group_norm.weight"], p[f"{a}.group_norm.bias"] = (c,), (c,)
    for k in ("to_q", "to_k", "to_v", "to_out.0"):
        p[f"{a}.{k}.weight"], p[f"{a}.{k}.bias"] = (c, c), (c,)
    _resnet_shapes(p, f"{name}.resnets.1", c, c)


def param_shapes(cfg):
    p = OrderedDict()
    boc, L, zc, ic = cfg["block_out_channels"], cfg["layers_per_block"], cfg["latent_channels"], cfg["in_channels"]
    p["encoder.conv_in.weight"], p["encoder.conv_in.bias"] = (boc[0], ic, 3, 3), (boc[0],)
    cin = boc[0]
    for i, c in enumerate(boc):
        for j in range(L):
            _resnet_shapes(p, f"encoder.down_blocks.{i}.resnets.{j}", cin, c)
            cin = c
        if i != len(boc) - 1:
            p[f"encoder.down_blocks.{i}.downsamplers.0.conv.weight"], p[f"encoder.down_blocks.{i}.downsamplers.0.conv.bias"] = (c, c, 3, 3), (c,)
    _mid_shapes(p, "encoder.mid_block", boc[-1])
    p["encoder.conv_norm_out.weight"], p["encoder.conv_norm_out.bias"] = (boc[-1],), (boc[-1],)
    p["encoder.conv_out.weight"], p["encoder.conv_out.bias"] = (2 * zc, boc[-1], 3, 3), (2 * zc,)
    p["quant_conv.weight"], p["quant_conv.bias"] = (2 * zc, 2 * zc, 1, 1), (2 * zc,)
    p["post_quant_conv.weight"], p["post_quant_conv.bias"] = (zc, zc, 1, 1), (zc,)
    rev = list(reversed(boc))
    p["decoder.conv_in.weight"], p["decoder.conv_in.bias"] = (rev[0], zc, 3, 3), (rev[0],)
    _mid_shapes(p, "decoder.mid_block", rev[0])
    cin = rev[0]
    for i, c in enumerate(rev):
        for j in range(L + 1):
            _resnet_shapes(p, f"decoder.up_blocks.{i}.resnets.{j}", cin, c)
            cin = c
        if i != len(rev) - 1:
            p[f"decoder.up_blocks.{i}.upsamplers.0.conv.weight"], p[f"decoder.up_blocks.{i}.upsamplers.0.conv.bias"] = (c, c, 3, 3), (c,)
    p["decoder.conv_norm_out.weight"], p["decoder.conv_norm_out.bias"] = (rev[-1],), (rev[-1],)
    p["decoder.conv_out.weight"], p["decoder.conv_out.bias"] = (ic, rev[-1], 3, 3), (ic,)
    return p


def init_state(cfg, seed=0):
    g = torch.Generator().manual_seed(seed)
    sd = OrderedDict()
    for n, shp in param_shapes(cfg).items():
        t = torch.randn(shp, generator=g)
        if len(shp) >= 2:
            t = t / math.sqrt(math.prod(shp[1:]))
        else:
            t = t * 0.02
        if "norm" in n and n.endswith(".weight"):
            t = 1.0 + t
        sd[n] = t
    return sd


def _gn(sd, name, x, silu):
    y = F.group_norm(x, 32, sd[name + ".weight"], sd[name + ".bias"], 1e-6)
    return F.silu(y) if silu else y


def _resnet(sd, name, x):
    h = F.conv2d(_gn(sd, name + ".norm1", x, True), sd[name + ".conv1.weight"], sd[name + ".conv1.bias"], padding=1)
    h = F.conv2d(_gn(sd, name + ".norm2", h, True), sd[name + ".conv2.weight"], sd[name + ".conv2.bias"], padding=1)
    if (name + ".conv_shortcut.weight") in sd:
        x = F.conv2d(x, sd[name + ".conv_shortcut.weight"], sd[name + ".conv_shortcut.bias"])
    return x + h


def _mid(sd, name, x):
    x = _resnet(sd, name + ".resnets.0", x)
    a = name + ".attentions.0"
    B, C, H, W = x.shape
    hN = _gn(sd, a + ".group_norm", x, False).reshape(B, C, H * W).transpose(1, 2)
    q = F.linear(hN, sd[a + ".to_q.weight"], sd[a + ".to_q.bias"])
    k = F.linear(hN, sd[a + ".to_k.weight"], sd[a + ".to_k.bias"])
    v = F.linear(hN, sd[a + ".to_v.weight"], sd[a + ".to_v.bias"])
    o = torch.softmax(q @ k.transpose(1, 2) / math.sqrt(C), dim=-1) @ v            # one head of width C
    o = F.linear(o, sd[a + ".to_out.0.weight"], sd[a + ".to_out.0.bias"])
    x = x + o.transpose(1, 2).reshape(B, C, H, W)
    return _resnet(sd, name + ".resnets.1", x)


def decode(cfg, sd, z):
    """AutoencoderKL.decode(z).sample: z [B, 4, h, w] (already divided by the scaling factor) -> image [B, 3, 8h, 8w]."""
    boc, L = cfg["block_out_channels"], cfg["layers_per_block"]
    x = F.conv2d(z, sd["post_quant_conv.weight"], sd["post_quant_conv.bias"])
    x = F.conv2d(x, sd["decoder.conv_in.weight"], sd["decoder.conv_in.bias"], padding=1)
    x = _mid(sd, "decoder.mid_block", x)
    for i in range(len(boc)):
        for j in range(L + 1):
            x = _resnet(sd, f"decoder.up_blocks.{i}.resnets.{j}", x)
        if i != len(boc) - 1:
            x = F.interpolate(x, scale_factor=2.0, mode="nearest")
            x = F.conv2d(x, sd[f"decoder.up_blocks.{i}.upsamplers.0.conv.weight"], sd[f"decoder.up_blocks.{i}.upsamplers.0.conv.bias"], padding=1)
    x = _gn(sd, "decoder.conv_norm_out", x, True)
    return F.conv2d(x, sd["decoder.conv_out.weight"], sd["decoder.conv_out.bias"], padding=1)


def encode_moments(cfg, sd, img):
    """AutoencoderKL.encode(img).latent_dist.parameters: img [B, 3, H, W] in [-1, 1] -> [B, 8, H/8, W/8] (mean | logvar)."""
    boc, L = cfg["block_out_channels"], cfg["layers_per_block"]
    x = F.conv2d(img, sd["encoder.conv_in.weight"], sd["encoder.conv_in.bias"], padding=1)
    for i in range(len(boc)):
        for j in range(L):
            x = _resnet(sd, f"encoder.down_blocks.{i}.resnets.{j}", x)
        if i != len(boc) - 1:
            x = F.pad(x, (0, 1, 0, 1))                     # diffusers Downsample2D(padding=0): pad right / bottom only
            x = F.conv2d(x, sd[f"encoder.down_blocks.{i}.downsamplers.0.conv.weight"], sd[f"encoder.down_blocks.{i}.downsamplers.0.conv.bias"], stride=2)
    x = _mid(sd, "encoder.mid_block", x)
    x = _gn(sd, "encoder.conv_norm_out", x, True)
    x = F.conv2d(x, sd["encoder.conv_out.weight"], sd["encoder.conv_out.bias"], padding=1)
    return F.conv2d(x, sd["quant_conv.weight"], sd["quant_conv.bias"])


def postprocess(img):
    """VaeImageProcessor.postprocess (denormalize): (img / 2 + 0.5).clamp(0, 1)."""
    return (img / 2 + 0.5).clamp(0, 1)
