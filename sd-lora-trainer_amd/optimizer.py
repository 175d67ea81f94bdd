"""The optimizer seam of the reference (trainer/optimizer.py:237-275 `OptimizerCollection`, main.py:271-291, 381-382):
objects with a mutable `.param_groups[0]['lr']`, `.step()` and `.zero_grad()`.

In the reference each is a torch.optim.AdamW; here both are views onto the fused device-side AdamW of `step.TrainStep`
(sdlt_adamw_fused): main.py-style code writes the learning rates into `param_groups`, `OptimizerCollection.step()`
hands them to the captured step.  Gradients are overwritten every step, so `zero_grad` is a no-op kept for the interface."""


class _FusedAdamWHandle:
    def __init__(self, name, lr, weight_decay, betas=(0.9, 0.999), eps=1e-8):
        self.name = name
        self.param_groups = [dict(lr=lr, weight_decay=weight_decay, betas=betas, eps=eps)]

    def zero_grad(self):
        pass


class OptimizerCollection:
    """order of the reference: textual inversion, text-encoder LoRA (not supported: off by default, config.py:115), unet."""

    def __init__(self, train_step, config):
        self.ts = train_step
        self.optimizers = {
            "textual_inversion": _FusedAdamWHandle("ti", config.ti_lr, config.ti_weight_decay) if train_step.ti is not None else None,
            "text_encoders": None,
            "unet": _FusedAdamWHandle("unet", 1e-4, config.lora_weight_decay),
        }
        self.learning_rate_tracker = {k: [] for k, v in self.optimizers.items() if v is not None}

    def step(self):
        """Runs forward, backward and both fused AdamW updates of ONE training step with the current learning rates."""
        lr_unet = self.optimizers["unet"].param_groups[0]["lr"]
        ti = self.optimizers["textual_inversion"]
        lr_ti = ti.param_groups[0]["lr"] if ti is not None else 0.0
        self.ts.run(lr_unet, lr_ti)
        for k in self.learning_rate_tracker:
            self.learning_rate_tracker[k].append(self.optimizers[k].param_groups[0]["lr"])

    def zero_grad(self):
        pass
