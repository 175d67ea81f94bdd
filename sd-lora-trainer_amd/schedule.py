"""Per-step host scalars of the reference's loop: learning-rate schedules (main.py:236-240, 265-291) and caption
dropout (main.py:300-304)."""
import numpy as np


def base_unet_lr(is_lora: bool, disable_ti: bool) -> float:
    """main.py:236-240: cold-start lr of the exponential warm-up."""
    if not is_lora:
        return 1.0e-5
    return 2.0e-4 if disable_ti else 5.0e-5


def completion_fraction(epoch, step_in_epoch, steps_per_epoch, num_train_epochs):
    """main.py:265-266."""
    return (epoch + step_in_epoch / steps_per_epoch) / num_train_epochs


def learning_rates(config, global_step, completion_f, ti_active=True, text_lora_active=False):
    """-> dict(unet=..., textual_inversion=..., text_encoders=...) exactly as main.py:268-291 writes them into
    param_groups[0]['lr'].  With ti_optimizer == 'prodigy' the loop leaves the TI lr untouched (main.py:269), i.e. at the
    1.0 the optimizer was built with (optimizer.py:130,137) - no decay and no freeze."""
    out = {}
    if ti_active:
        if config.ti_optimizer == "prodigy":
            lr_ti = 1.0
        else:
            lr_ti = config.ti_lr * (1 - completion_f) ** 1.7
            if completion_f > config.freeze_ti_after_completion_f:
                lr_ti = 0.0
        out["textual_inversion"] = lr_ti
    if text_lora_active:
        lr = config.text_encoder_lora_lr * (1 - completion_f) ** 2.0
        if config.txt_encoders_lr_warmup_steps > 0:
            lr *= min(global_step / config.txt_encoders_lr_warmup_steps, 1.0)
        out["text_encoders"] = lr
    base = base_unet_lr(config.is_lora, config.disable_ti)
    lr_unet = base * (config.unet_lr / base) ** (global_step / config.unet_lr_warmup_steps)
    if completion_f < config.freeze_unet_before_completion_f:
        lr_unet = 0.0
    out["unet"] = lr_unet
    return out


def apply_caption_dropout(captions, caption_dropout, tok_string, rng=np.random):
    """main.py:300-304: with probability caption_dropout a caption becomes the bare trigger string (token_dict['TOK']).
    Uses numpy's global RNG like the reference (seeded by seed_everything)."""
    captions = list(captions)
    if caption_dropout > 0.0:
        for i in range(len(captions)):
            if rng.rand() < caption_dropout:
                captions[i] = tok_string
    return captions
