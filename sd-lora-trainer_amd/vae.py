"""AutoencoderKL (the SD1.5 / SDXL VAE) as forward plans over the HIP kernels of the training step - NHWC bf16 activations,
3x3 convs as implicit GEMMs, GroupNorm(+SiLU) fused - for the two places the reference runs it either side of the step:
  * `vae.decode(latents / scaling_factor)` at the end of the validation render (trainer/inference.py:289-385), and
  * `vae.encode(image).latent_dist` once per training image (trainer/dataset.py:141-157); the moments feed
    dataset.LatentCache, which re-samples the posterior on every fetch.
Weights use the diffusers AutoencoderKL state-dict names.  The mid-block attention is one head as wide as the block (512):
the scores of an image are a dense [N, N] GEMM, the row softmax is a torch op between the two GEMMs (N = 16384 for a 1024 px
image: 1 GB of fp32 scores, trivial next to 288 GB).  The encoder's stride-2 convs pad right/bottom only (diffusers
Downsample2D(padding=0)): computed as the pad-1 stride-1 conv sampled at the odd pixels, which reads exactly those taps.
"""
import math

import torch

from .unet import F32, Conv3x3, GroupNorm, Linear, _Module, _pad_to


class _Renamed(dict):
    def __init__(self, sd, extra):
        self.sd, self.extra = sd, extra

    def __getitem__(self, k):
        return self.extra[k] if k in self.extra else self.sd[k]

    def get(self, k, default=None):
        return self.extra.get(k, self.sd.get(k, default))

    def __contains__(self, k):
        return k in self.extra or k in self.sd


class VaeResnet(_Module):
    def __init__(self, rt, name, sd):
        super().__init__(rt, name)
        self.norm1 = GroupNorm(rt, name + ".norm1", sd, 1e-6, silu=True)
        self.conv1 = Conv3x3(rt, name + ".conv1", sd, need_dx=False)
        self.norm2 = GroupNorm(rt, name + ".norm2", sd, 1e-6, silu=True)
        self.conv2 = Conv3x3(rt, name + ".conv2", sd, need_dx=False)
        self.shortcut = Linear(rt, name + ".conv_shortcut", sd, need_dx=False) if (name + ".conv_shortcut.weight") in sd else None

    def forward(self, x, B, H, W):
        h = self.conv1.forward(self.norm1.forward(x, None, B, H * W), B, H, W, train=False)
        sc = x if self.shortcut is None else self.shortcut.forward(x, train=False)
        return self.conv2.forward(self.norm2.forward(h, None, B, H * W), B, H, W, residual=sc, train=False)


class VaeAttention(_Module):
    def __init__(self, rt, name, sd):
        super().__init__(rt, name)
        self.norm = GroupNorm(rt, name + ".group_norm", sd, 1e-6, silu=False)
        self.q = Linear(rt, name + ".to_q", sd, need_dx=False)
        self.k = Linear(rt, name + ".to_k", sd, need_dx=False)
        self.v = Linear(rt, name + ".to_v", sd, need_dx=False)
        self.o = Linear(rt, name + ".to_out.0", sd, need_dx=False)
        self.C = self.q.N

    def forward(self, x, B, N):
        rt, C = self.rt, self.C
        hn = self.norm.forward(x, None, B, N)
        q, k = self.q.forward(hn, train=False), self.k.forward(hn, train=False)
        Np = _pad_to(N, 64)
        vt = self.buf("vt", C, B * Np, zero=True)
        for b in range(B):                       # V^T per image (the GEMM's transposed side output), keys padded with zeros
            self.v.forward(hn[b * N:(b + 1) * N], Ct=vt[:, b * Np:(b + 1) * Np], key=("v", b), train=False)
        O = self.buf("O", B * N, C)
        S = self.buf("S", N, Np, dtype=F32, zero=True)
        P = self.buf("P", N, Np, zero=True)
        for b in range(B):
            rt.ops.gemm(q[b * N:(b + 1) * N], k[b * N:(b + 1) * N], S[:, :N], alpha=1.0 / math.sqrt(C))
            P[:, :N] = torch.softmax(S[:, :N], dim=-1).to(P.dtype)
            rt.ops.gemm(P, vt[:, b * Np:(b + 1) * Np], O[b * N:(b + 1) * N])
        return self.o.forward(O, residual=x, train=False)


class _Mid(_Module):
    def __init__(self, rt, name, sd):
        super().__init__(rt, name)
        self.r0 = VaeResnet(rt, name + ".resnets.0", sd)
        self.attn = VaeAttention(rt, name + ".attentions.0", sd)
        self.r1 = VaeResnet(rt, name + ".resnets.1", sd)

    def forward(self, x, B, H, W):
        return self.r1.forward(self.attn.forward(self.r0.forward(x, B, H, W), B, H * W), B, H, W)


class VaeDecoder(_Module):
    """decode(z [B, 4, h, w] fp32, already / scaling_factor) -> image [B, 3, 8h, 8w] fp32 (before postprocess)."""

    def __init__(self, rt, sd, n_levels=None, layers_per_block=None):
        super().__init__(rt, "vae.decoder")
        nl = 0
        while f"decoder.up_blocks.{nl}.resnets.0.norm1.weight" in sd:
            nl += 1
        L = 0
        while f"decoder.up_blocks.0.resnets.{L}.norm1.weight" in sd:
            L += 1
        self.pq_w = sd["post_quant_conv.weight"].to(rt.device, F32).reshape(sd["post_quant_conv.weight"].shape[0], -1)
        self.pq_b = sd["post_quant_conv.bias"].to(rt.device, F32)
        self.zc = self.pq_w.shape[0]
        self.conv_in = Conv3x3(rt, "decoder.conv_in", sd, cin_pad=64, need_dx=False)
        self.mid = _Mid(rt, "decoder.mid_block", sd)
        self.ups = []
        for i in range(nl):
            res = [VaeResnet(rt, f"decoder.up_blocks.{i}.resnets.{j}", sd) for j in range(L)]
            up = Conv3x3(rt, f"decoder.up_blocks.{i}.upsamplers.0.conv", sd, ups=2, need_dx=False) if i != nl - 1 else None
            self.ups.append((res, up))
        self.norm_out = GroupNorm(rt, "decoder.conv_norm_out", sd, 1e-6, silu=True)
        # conv_out has 3 output channels: padded with a zero row to 4 so that the fp32 output rows stay 16-byte aligned
        w, b = sd["decoder.conv_out.weight"], sd["decoder.conv_out.bias"]
        self.ic = w.shape[0]
        pad = _pad_to(self.ic, 4) - self.ic
        extra = {"decoder.conv_out.weight": torch.cat([w, w.new_zeros(pad, *w.shape[1:])]), "decoder.conv_out.bias": torch.cat([b, b.new_zeros(pad)])}
        self.conv_out = Conv3x3(rt, "decoder.conv_out", _Renamed(sd, extra), need_dx=False)

    @torch.no_grad()
    def decode(self, z):
        rt = self.rt
        B, zc, h, w = z.shape
        z = torch.einsum("oc,bchw->bohw", self.pq_w, z.to(rt.device, F32)) + self.pq_b.view(1, -1, 1, 1)     # post_quant_conv (1x1, 4 -> 4)
        x64 = self.buf("x64", B * h * w, 64, zero=True)
        x64[:, :zc] = z.permute(0, 2, 3, 1).reshape(B * h * w, zc).to(x64.dtype)
        x = self.conv_in.forward(x64, B, h, w, train=False)
        x = self.mid.forward(x, B, h, w)
        ch, cw = h, w
        for res, up in self.ups:
            for r in res:
                x = r.forward(x, B, ch, cw)
            if up is not None:
                x = up.forward(x, B, ch, cw, train=False)
                ch, cw = ch * 2, cw * 2
        hn = self.norm_out.forward(x, None, B, ch * cw)
        out = self.conv_out.forward(hn, B, ch, cw, out=self.buf("img", B * ch * cw, _pad_to(self.ic, 4), dtype=F32), train=False)
        return out.view(B, ch, cw, -1)[..., : self.ic].permute(0, 3, 1, 2).contiguous()


class VaeEncoder(_Module):
    """encode_moments(img [B, 3, H, W] fp32 in [-1, 1]) -> [B, 8, H/8, W/8] fp32 (mean | logvar) = latent_dist.parameters."""

    def __init__(self, rt, sd):
        super().__init__(rt, "vae.encoder")
        nl = 0
        while f"encoder.down_blocks.{nl}.resnets.0.norm1.weight" in sd:
            nl += 1
        L = 0
        while f"encoder.down_blocks.0.resnets.{L}.norm1.weight" in sd:
            L += 1
        self.conv_in = Conv3x3(rt, "encoder.conv_in", sd, cin_pad=64, need_dx=False)
        self.ic = self.conv_in.Cin
        self.downs = []
        for i in range(nl):
            res = [VaeResnet(rt, f"encoder.down_blocks.{i}.resnets.{j}", sd) for j in range(L)]
            ds = Conv3x3(rt, f"encoder.down_blocks.{i}.downsamplers.0.conv", sd, need_dx=False) if i != nl - 1 else None
            self.downs.append((res, ds))
        self.mid = _Mid(rt, "encoder.mid_block", sd)
        self.norm_out = GroupNorm(rt, "encoder.conv_norm_out", sd, 1e-6, silu=True)
        self.conv_out = Conv3x3(rt, "encoder.conv_out", sd, need_dx=False)
        self.q_w = sd["quant_conv.weight"].to(rt.device, F32).reshape(sd["quant_conv.weight"].shape[0], -1)
        self.q_b = sd["quant_conv.bias"].to(rt.device, F32)

    @torch.no_grad()
    def encode_moments(self, img):
        rt = self.rt
        B, ic, H, W = img.shape
        x64 = self.buf("x64", B * H * W, 64, zero=True)
        x64[:, :ic] = img.to(rt.device, F32).permute(0, 2, 3, 1).reshape(B * H * W, ic).to(x64.dtype)
        x = self.conv_in.forward(x64, B, H, W, train=False)
        ch, cw = H, W
        for i, (res, ds) in enumerate(self.downs):
            for r in res:
                x = r.forward(x, B, ch, cw)
            if ds is not None:
                # pad (0,1,0,1) + stride 2 + no padding == the pad-1 stride-1 conv sampled at the odd pixels
                full = ds.forward(x, B, ch, cw, train=False)
                C = full.shape[1]
                x = self.buf(("ds", i), B * (ch // 2) * (cw // 2), C)
                x.view(B, ch // 2, cw // 2, C).copy_(full.view(B, ch, cw, C)[:, 1::2, 1::2])
                ch, cw = ch // 2, cw // 2
        x = self.mid.forward(x, B, ch, cw)
        hn = self.norm_out.forward(x, None, B, ch * cw)
        m = self.conv_out.forward(hn, B, ch, cw, out=self.buf("mom", B * ch * cw, self.conv_out.Cout, dtype=F32), train=False)
        m = m.view(B, ch, cw, -1).permute(0, 3, 1, 2)
        return torch.einsum("oc,bchw->bohw", self.q_w, m) + self.q_b.view(1, -1, 1, 1)                      # quant_conv (1x1, 8 -> 8)


def postprocess(img):
    """VaeImageProcessor.postprocess: denormalise to [0, 1] (the reference then converts to PIL and saves JPEG q95)."""
    return (img / 2 + 0.5).clamp(0, 1)
