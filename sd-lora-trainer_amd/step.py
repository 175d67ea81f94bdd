"""One optimisation step of the reference's training loop (/root/reference main.py:263-382) as a static plan
over the HIP kernels: add-noise -> UNet(+LoRA) forward -> masked/SNR-weighted MSE -> explicit backward ->
grouped LoRA gradients -> fused AdamW(+L1) -> bf16 shadow refresh.  On the GPU the whole body is captured
once into a hipGraph (torch.cuda.CUDAGraph is hipGraph on ROCm) and replayed; per-step scalars (learning
rates, bias corrections) live in a small device buffer that is updated before each replay.
"""
import math

import torch

from . import ops as _ops
from .unet import CTX_PAD, F32, Runtime, UNet


def ddpm_alphas_cumprod(n=1000, beta_start=0.00085, beta_end=0.012):
    """diffusers DDPMScheduler(beta_schedule="scaled_linear") table used by add_noise / compute_snr."""
    betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, n, dtype=torch.float32) ** 2
    return torch.cumprod(1.0 - betas, dim=0)


class TrainStep:
    def __init__(self, rt: Runtime, unet: UNet, *, latent_hw, snr_gamma=5.0, v_prediction=False, l1_penalty=0.03,
                 weight_decay=0.004, grad_accum=1, betas=(0.9, 0.999), eps=1e-8):
        self.rt, self.unet = rt, unet
        B, (h, w) = rt.B, latent_hw
        cfg = unet.cfg
        self.B, self.h, self.w = B, h, w
        self.snr_gamma, self.v_pred, self.l1_penalty, self.wd, self.grad_accum = snr_gamma, v_prediction, l1_penalty, weight_decay, grad_accum
        self.betas, self.eps = betas, eps
        dev = rt.device
        z = lambda *s, dtype=F32: torch.zeros(*s, dtype=dtype, device=dev)  # noqa: E731
        self.latent, self.noise, self.mask = z(B, 4, h, w), z(B, 4, h, w), z(B, 4, h, w)
        self.noisy = z(B, 4, h, w)
        self.timesteps = z(B, dtype=torch.int64)
        self.timesteps_f = z(B)
        self.ctx = rt.zeros(B * CTX_PAD, cfg["cross_dim"])
        self.dctx = rt.zeros(B * CTX_PAD, cfg["cross_dim"])
        self.pooled = self.time_ids = None
        if cfg["addition"]:
            self.pooled = rt.zeros(B, cfg["proj_class_in"] - 6 * cfg["addition_time_embed_dim"])
            self.time_ids = z(B * 6)
        self.acp = ddpm_alphas_cumprod().to(dev)
        self.x64 = rt.zeros(B * h * w, 64)
        self.dpred64 = rt.zeros(B * h * w, 64)
        self.sums, self.loss, self.l1_sum = z(B * 2), z(1), z(1)
        self.hyper = z(16)
        self.opt_step = 0
        self.graph = None

    # -------------------------------------------------------------------------------- inputs
    def set_batch(self, latent, noise, timesteps, mask, ctx, pooled=None, time_ids=None):
        """latent/noise/mask [B,4,h,w] fp32, timesteps int64 [B], ctx [B,77,D]; SDXL: pooled [B,P], time_ids [B,6]."""
        self.latent.copy_(latent)
        self.noise.copy_(noise)
        self.mask.copy_(mask)
        self.timesteps.copy_(timesteps)
        self.timesteps_f.copy_(timesteps.to(torch.float32))
        self.ctx.view(self.B, CTX_PAD, -1)[:, :77].copy_(ctx)
        if self.pooled is not None:
            self.pooled.copy_(pooled)
            self.time_ids.copy_(time_ids.reshape(-1).to(torch.float32))

    def set_hyper(self, lr):
        """Host scalars of this optimiser step -> device buffer (see sdlt_adamw_fused)."""
        self.opt_step += 1
        b1, b2 = self.betas
        n = self.unet.arena.n
        vals = [lr, b1, b2, self.eps, self.wd, 1.0 - b1 ** self.opt_step, 1.0 - b2 ** self.opt_step,
                self.l1_penalty / n, 1.0]
        self.hyper[: len(vals)].copy_(torch.tensor(vals, dtype=torch.float32))

    # -------------------------------------------------------------------------------- the step body
    def forward_backward(self):
        rt, u = self.rt, self.unet
        ops = rt.ops
        ops.add_noise_nhwc(self.latent, self.noise, self.timesteps, self.acp, self.x64, self.noisy)
        pred = u.forward(self.x64, self.timesteps_f, self.ctx, self.pooled, self.time_ids, B=self.B, H=self.h, W=self.w)
        ops.masked_mse_fwd_bwd(pred, self.noise, self.noisy, self.mask, self.timesteps, self.acp, self.sums, self.loss,
                               self.dpred64, snr_gamma=self.snr_gamma, v_prediction=self.v_pred, loss_scale=1.0 / self.grad_accum)
        self.dctx.zero_()
        u.backward(self.dpred64, self.dctx)
        return pred

    def optimizer_step(self):
        a = self.unet.arena
        self.rt.ops.adamw_fused(a.params, a.grads, a.m, a.v, self.hyper, self.l1_sum)
        a.refresh_shadows()

    def body(self):
        self.forward_backward()
        self.optimizer_step()

    # -------------------------------------------------------------------------------- graph capture / replay
    def capture(self, warmup=2):
        """Runs the body eagerly `warmup` times (allocates every persistent buffer, builds the grouped-gradient
        plan), then captures it.  AdamW state is restored afterwards so capture does not count as training."""
        a = self.unet.arena
        snap = [t.clone() for t in (a.params, a.m, a.v)]
        step0 = self.opt_step
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(warmup):
                self.body()
        torch.cuda.current_stream().wait_stream(s)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.body()
        for t, c in zip((a.params, a.m, a.v), snap):
            t.copy_(c)
        a.refresh_shadows()
        self.opt_step = step0

    def run(self, lr):
        self.set_hyper(lr)
        if self.graph is not None:
            self.graph.replay()
        else:
            self.body()

    def total_loss(self):
        """img loss + L1 penalty as the reference logs it (main.py:339-361); forces a device sync."""
        return float(self.loss) + self.l1_penalty * float(self.l1_sum) / self.unet.arena.n
