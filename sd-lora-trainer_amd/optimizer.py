"""The optimizer seam of the reference (trainer/optimizer.py:237-275 `OptimizerCollection`, main.py:271-291, 381-382):
objects with a mutable `.param_groups[0]['lr']`, `.step()` and `.zero_grad()`.

In the reference each is a torch.optim.AdamW; here both are views onto the fused device-side AdamW of `step.TrainStep`
(sdlt_adamw_fused): main.py-style code writes the learning rates into `param_groups`, `OptimizerCollection.step()`
hands them to the captured step.  Gradients are overwritten every step, so `zero_grad` is a no-op kept for the interface.
`unet_optimizer_type` / `ti_optimizer` = "prodigy" select the device-side Prodigy step (sdlt_prodigy_step) instead."""


class _FusedAdamWHandle:
    def __init__(self, name, lr, weight_decay, betas=(0.9, 0.999), eps=1e-8):
        self.name = name
        self.param_groups = [dict(lr=lr, weight_decay=weight_decay, betas=betas, eps=eps)]

    def zero_grad(self):
        pass


class _FusedProdigyHandle:
    """Prodigy group (prodigyopt 1.0 as built at trainer/optimizer.py:24-34 / 135-145) backed by step.ProdigyState.
    `param_groups[0]` carries the keys `get_current_lr` reads (d, k, betas, use_bias_correction, optimizer.py:214-223);
    d and k live on the device and are pulled by `sync()`."""

    def __init__(self, name, state, lr, weight_decay):
        self.name, self.state = name, state
        self.param_groups = [dict(state.group(lr), weight_decay=weight_decay)]

    def sync(self):
        self.param_groups[0].update(self.state.group(self.param_groups[0]["lr"]))

    def zero_grad(self):
        pass


def get_current_lr(optimizer):
    """trainer/optimizer.py:206-234: the effective step size d * lr * bias_correction of a Prodigy group, the plain lr of
    anything else."""
    g = optimizer.param_groups[0]
    if "d" not in g:
        return g["lr"]
    optimizer.sync()
    bc = 1.0
    if g["use_bias_correction"]:
        b1, b2 = g["betas"]
        bc = ((1 - b2 ** (g["k"] + 1)) ** 0.5) / (1 - b1 ** (g["k"] + 1))
    return g["d"] * g["lr"] * bc


class OptimizerCollection:
    """order of the reference: textual inversion, text-encoder LoRA (off by default, config.py:115), unet."""

    def __init__(self, train_step, config):
        self.ts = train_step
        self.optimizers = {
            "textual_inversion": None,
            "text_encoders": None,
            "unet": _FusedAdamWHandle("unet", 1e-4, config.lora_weight_decay) if train_step.prodigy is None
            else _FusedProdigyHandle("unet", train_step.prodigy, 1.0, config.lora_weight_decay),
        }
        if train_step.ti is not None:
            self.optimizers["textual_inversion"] = _FusedAdamWHandle("ti", config.ti_lr, config.ti_weight_decay) \
                if train_step.prodigy_ti is None else _FusedProdigyHandle("ti", train_step.prodigy_ti, 1.0, config.ti_weight_decay)
        if getattr(train_step, "te_arena", None) is not None:      # optimizer.py:190-199: AdamW only
            self.optimizers["text_encoders"] = _FusedAdamWHandle("text_encoders", config.text_encoder_lora_lr, config.text_encoder_lora_weight_decay)
        self.learning_rate_tracker = {k: [] for k, v in self.optimizers.items() if v is not None}

    def step(self, last_batch=False):
        """Runs forward, backward and the fused optimizer updates of ONE training step with the current learning rates (under
        gradient accumulation the optimizers only step on every k-th call or on the last batch of an epoch, main.py:366)."""
        lr_unet = self.optimizers["unet"].param_groups[0]["lr"]
        ti = self.optimizers["textual_inversion"]
        lr_ti = ti.param_groups[0]["lr"] if ti is not None else 0.0
        te = self.optimizers["text_encoders"]
        self.ts.run(lr_unet, lr_ti, te.param_groups[0]["lr"] if te is not None else 0.0, last_batch=last_batch)
        for k in self.learning_rate_tracker:
            self.learning_rate_tracker[k].append(self.optimizers[k].param_groups[0]["lr"])

    def zero_grad(self):
        pass
