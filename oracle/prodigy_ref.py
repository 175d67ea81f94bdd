"""ORACLE (test infrastructure only - never imported by the product path).

CPU restatement of Prodigy as the reference configures it (trainer/optimizer.py:24-34 for the UNet LoRA tensors,
:135-145 for the token-embedding tables; effective learning-rate read-out trainer/optimizer.py:206-234).

PARITY UNPINNED: the algorithm lives in the third-party package `prodigyopt==1.0` (requirements.txt:18), which is not in
/root/reference and not installed here, and the reference holds no test or golden vector for it.  This file restates the
published algorithm of that release (Mishchenko & Defazio, "Prodigy: An Expeditiously Adaptive Parameter-Free Learner",
Algorithm 4 / the Adam variant with the package's bias-correction, safeguard-warmup, growth-rate and decoupled-decay
options), following the package's order of operations:

    k, d, d0, d_max, d_numerator carried in the (single) parameter group; beta3 = sqrt(beta2) by default
    bias_correction = sqrt(1 - beta2^(k+1)) / (1 - beta1^(k+1))   if use_bias_correction else 1
    dlr = d * lr * bias_correction
    d_numerator *= beta3
    per tensor (only while lr > 0):
        [coupled decay: g += wd * p]
        d_numerator += (d / d0) * dlr * <g, p0 - p>
        exp_avg     = beta1 exp_avg    + d (1 - beta1) g
        exp_avg_sq  = beta2 exp_avg_sq + d^2 (1 - beta2) g^2
        s           = beta3 s + (d / d0) * (d if safeguard_warmup else dlr) * g
        d_denom    += sum |s|
    if d_denom == 0: return                                  (nothing is written back, k does not advance)
    if lr > 0: d_hat = d_coef d_numerator / d_denom ; if d == d0: d = max(d, d_hat)
               d_max = max(d_max, d_hat) ; d = min(d_max, d * growth_rate)
    per tensor: denom = sqrt(exp_avg_sq) + d eps             (the NEW d)
                [decoupled decay: p -= wd * dlr * p]         (dlr of the OLD d)
                p -= dlr * exp_avg / denom
    k += 1

What IS pinned (tests/test_prodigy_cpu.py): closed-form first step, the growth-rate clamp, the lr == 0 no-op, scale
invariance of the iterates to a rescaling of the loss (the property the method is named for), and the equivalence of
stepping whole embedding tables with masked gradients and stepping the trainable rows only.
"""
import math

import torch


class Prodigy:
    """Same constructor keywords as prodigyopt.Prodigy; `params` is a list of fp32 tensors, gradients are passed to step()."""

    def __init__(self, params, lr=1.0, betas=(0.9, 0.999), beta3=None, eps=1e-8, weight_decay=0.0, decouple=True,
                 use_bias_correction=False, safeguard_warmup=False, d0=1e-6, d_coef=1.0, growth_rate=float("inf")):
        self.params = list(params)
        self.param_groups = [dict(params=self.params, lr=lr, betas=betas, beta3=beta3, eps=eps, weight_decay=weight_decay,
                                  d=d0, d0=d0, d_max=d0, d_numerator=0.0, d_coef=d_coef, k=0, growth_rate=growth_rate,
                                  use_bias_correction=use_bias_correction, decouple=decouple, safeguard_warmup=safeguard_warmup)]
        self.state = [None] * len(self.params)

    def step(self, grads):
        g_ = self.param_groups[0]
        beta1, beta2 = g_["betas"]
        beta3 = g_["beta3"] if g_["beta3"] is not None else math.sqrt(beta2)
        k, d, d0, d_max, lr = g_["k"], g_["d"], g_["d0"], g_["d_max"], g_["lr"]
        bias_correction = (math.sqrt(1 - beta2 ** (k + 1)) / (1 - beta1 ** (k + 1))) if g_["use_bias_correction"] else 1.0
        dlr = d * lr * bias_correction
        d_numerator = g_["d_numerator"] * beta3
        d_denom = 0.0
        decay, decouple = g_["weight_decay"], g_["decouple"]
        for i, (p, grad) in enumerate(zip(self.params, grads)):
            grad = grad.clone()
            if decay != 0 and not decouple:
                grad.add_(p, alpha=decay)
            if self.state[i] is None:
                self.state[i] = dict(s=torch.zeros_like(p), p0=p.clone(), exp_avg=torch.zeros_like(p), exp_avg_sq=torch.zeros_like(p))
            st = self.state[i]
            if lr > 0.0:
                d_numerator += (d / d0) * dlr * torch.dot(grad.flatten().double(), (st["p0"] - p).flatten().double()).item()
                st["exp_avg"].mul_(beta1).add_(grad, alpha=d * (1 - beta1))
                st["exp_avg_sq"].mul_(beta2).addcmul_(grad, grad, value=d * d * (1 - beta2))
                st["s"].mul_(beta3).add_(grad, alpha=(d / d0) * (d if g_["safeguard_warmup"] else dlr))
                d_denom += st["s"].abs().double().sum().item()
        if d_denom == 0:
            return
        d_hat = d
        if lr > 0.0:
            d_hat = g_["d_coef"] * d_numerator / d_denom
            if d == d0:
                d = max(d, d_hat)
            d_max = max(d_max, d_hat)
            d = min(d_max, d * g_["growth_rate"])
        g_.update(d_numerator=d_numerator, d_denom=d_denom, d=d, d_max=d_max, d_hat=d_hat)
        for p, st in zip(self.params, self.state):
            denom = st["exp_avg_sq"].sqrt().add_(d * g_["eps"])
            if decay != 0 and decouple:
                p.add_(p, alpha=-decay * dlr)
            p.addcdiv_(st["exp_avg"], denom, value=-dlr)
        g_["k"] = k + 1


def effective_lr(group):
    """trainer/optimizer.py:206-234 (`get_current_lr`) for one Prodigy group: d * lr * bias_correction."""
    bc = 1.0
    if group["use_bias_correction"]:
        b1, b2 = group["betas"]
        bc = math.sqrt(1 - b2 ** (group["k"] + 1)) / (1 - b1 ** (group["k"] + 1))
    return group["d"] * group["lr"] * bc
