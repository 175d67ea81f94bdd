"""`TrainingConfig`: the reference's train_configs/*.json interface (/root/reference trainer/config.py:38-177).  Field names,
defaults, derived values and the JSON round trip are a CONTRACT (existing config files must drive this engine unchanged) and
are pinned against the reference on all of its shipped configs by tests/golden/config_snapshots.json; the model itself is
generated from the table below, which is organised by the stage of THIS engine that consumes each field.

Deliberate differences: the device is NOT chosen by probing free memory after torch is initialised
(trainer/utils/utils.py:64-89, racy for parallel launches - SURVEY.md App. C14); the launcher pins one GPU per process through
HIP_VISIBLE_DEVICES (parallel.py) and a job always uses cuda:0.  `pretrained_model` may also name a synthetic random-init model
("synthetic:sdxl" / "synthetic:sd15" / "synthetic:tinyxl" ...) because this environment has no network.
"""
import json
import os
import time
from datetime import datetime
from typing import List, Literal, Optional, Union

from pydantic import ConfigDict, create_model

REQUIRED = ...
Opt = Optional

# trainer/config.py:33-36 (paths are resolved by the host application; the version is what matters here)
pretrained_models = {v: {"path": f"models/checkpoints/{f}.safetensors", "url": None, "version": v}
                     for v, f in (("sdxl", "juggernautXL_v6"), ("sd15", "juggernaut_reborn"))}

_UNSAFE = ["\\", "/", ":", "*", "?", '"', "<", ">", "|", " ", "\n", "\t", "."]


def remove_delimiter_characters(name: str) -> str:
    """trainer/checkpoint.py:58-81: make a run name safe for file names."""
    if name is None:
        return name
    for ch in _UNSAFE:
        name = name.replace(ch, "_")
    return name


# field -> (annotation, default); REQUIRED = no default
FIELDS = {
    # ---- job identity, data source, model choice (train.build_models, dataset.LatentCache)
    "lora_training_urls": (str, REQUIRED), "concept_mode": (Literal["face", "style", "object"], REQUIRED),
    "name": (Opt[str], None), "output_dir": (str, "eden_lora_training_runs"), "seed": (Union[int, None], None),
    "sd_model_version": (Opt[Literal["sdxl", "sd15"]], None), "ckpt_path": (Opt[str], None), "pretrained_model": (Opt[dict], None),
    "weight_type": (Literal["fp16", "bf16", "fp32"], "bf16"), "device": (str, "cuda:0"), "allow_tf32": (bool, True),
    "dataloader_num_workers": (int, 0), "debug": (bool, False), "training_attributes": (dict, {}), "start_time": (float, 0.0), "job_time": (float, 0.0),
    # ---- step geometry and length (train.train, step.TrainStep)
    "resolution": (int, 512), "train_img_size": (Opt[List[int]], None), "train_aspect_ratio": (Opt[float], None), "aspect_ratio_bucketing": (bool, False),
    "train_batch_size": (int, 4), "gradient_accumulation_steps": (int, 1), "max_train_steps": (int, 300), "num_train_epochs": (Opt[int], None),
    "checkpointing_steps": (int, 10000), "noise_offset": (float, 0.02), "snr_gamma": (Opt[float], 5.0), "caption_dropout": (float, 0.1),
    # ---- what is trained: LoRA (unet.LoraArena) or every weight (fullft.WeightTrainer), and its optimizer (sdlt_adamw_fused / sdlt_prodigy_step)
    "is_lora": (bool, True), "lora_rank": (int, 16), "lora_alpha_multiplier": (float, 1.0), "use_dora": (bool, False),
    "unet_optimizer_type": (Literal["adamw", "prodigy", "AdamW8bit"], "adamw"), "unet_lr": (float, 0.0003), "unet_lr_warmup_steps": (Opt[int], None),
    "lora_weight_decay": (float, 0.004), "l1_penalty": (float, 0.03), "prodigy_d_coef": (float, 1.0), "unet_prodigy_growth_factor": (float, 1.05),
    "freeze_unet_before_completion_f": (float, 0.0),
    # ---- textual inversion (ti.TiState, step.TextStack) and its losses (daam.TokenAttentionLoss, sdlt_ti_std_reg)
    "disable_ti": (bool, False), "n_tokens": (int, 3), "inserting_list_tokens": (List[str], ["<s0>", "<s1>", "<s2>"]),
    "token_dict": (dict, {"TOK": "<s0><s1><s2>"}), "ti_optimizer": (Literal["adamw", "prodigy"], "adamw"), "ti_lr": (float, 0.001),
    "ti_weight_decay": (float, 0.0), "freeze_ti_after_completion_f": (float, 0.7), "token_warmup_steps": (int, 0),
    "token_attention_loss_w": (float, 3e-7), "cond_reg_w": (float, 0.0), "tok_cond_reg_w": (float, 0.0), "tok_cov_reg_w": (float, 0.0),
    # ---- text-encoder LoRA (clip.ClipTextEncoder(arena=))
    "text_encoder_lora_optimizer": (Union[None, Literal["adamw"]], None), "text_encoder_lora_lr": (float, 1.0e-5),
    "txt_encoders_lr_warmup_steps": (int, 200), "text_encoder_lora_weight_decay": (float, 1.0e-5), "text_encoder_lora_rank": (int, 16),
    # ---- validation render (sampler.render_images)
    "n_sample_imgs": (int, 4), "sample_imgs_lora_scale": (Opt[float], None), "validation_img_size": (Opt[Union[int, List[int]]], None),
    "prompt_modifier": (Opt[str], None),
    # ---- preprocessing stage (outside this engine; carried so that config files round-trip)
    "caption_prefix": (str, ""), "caption_model": (Literal["gpt4-v", "blip", "florence", "no_caption"], "florence"), "skip_gpt_cleanup": (bool, False),
    "left_right_flip_augmentation": (bool, True), "augment_imgs_up_to_n": (int, 40), "mask_target_prompts": (Union[None, str], None),
    "crop_based_on_salience": (bool, True), "use_face_detection_instead": (bool, False), "clipseg_temperature": (float, 0.5),
}

# unknown keys are silently ignored, like the reference's callers rely on
_Fields = create_model("_Fields", __config__=ConfigDict(extra="ignore"), **FIELDS)


def _derive(c, make_dirs):
    """Derived values of trainer/config.py:121-165, in the reference's order (the run directory name embeds three of them)."""
    if c.ckpt_path:
        c.pretrained_model = {"path": c.ckpt_path, "url": None, "version": None}
    elif c.pretrained_model is None:
        c.pretrained_model = pretrained_models.get(c.sd_model_version)
    c.name = remove_delimiter_characters(c.name or os.path.basename(c.lora_training_urls)[:40])
    stamp = datetime.now().strftime("%d%b_%H%M")
    c.output_dir = f"{c.output_dir}/{c.name}_{stamp}-{c.concept_mode}_res{c.resolution}_{c.max_train_steps}steps"
    if make_dirs:
        os.makedirs(c.output_dir, exist_ok=True)
    c.seed = int(time.time()) if c.seed is None else c.seed
    c.unet_lr_warmup_steps = c.max_train_steps if c.unet_lr_warmup_steps is None else c.unet_lr_warmup_steps
    c.checkpointing_steps = c.max_train_steps if c.checkpointing_steps < 1 else c.checkpointing_steps
    if c.concept_mode == "face":          # faces are not mirror-symmetric; the mask prompt is fixed
        c.left_right_flip_augmentation, c.mask_target_prompts = False, "face"
    if c.use_dora:                        # DoRA runs without the sparsity / decay terms
        c.l1_penalty = c.lora_weight_decay = c.text_encoder_lora_weight_decay = 0.0
    c.inserting_list_tokens = [f"<s{i}>" for i in range(c.n_tokens)]
    c.token_dict = {"TOK": "".join(c.inserting_list_tokens)}
    c.device = "cuda:0"                   # one visible device per process (parallel.job_env)
    c.start_time = time.time()


class TrainingConfig(_Fields):
    def __init__(self, **data):
        make_dirs = data.pop("_make_dirs", True)
        super().__init__(**data)
        _derive(self, make_dirs)

    @classmethod
    def from_json(cls, file_path: str, **overrides):
        with open(file_path, "r") as f:
            data = json.load(f)
        data.update(overrides)
        return cls(**data)

    def save_as_json(self, file_path: str) -> None:
        with open(file_path, "w") as f:
            json.dump(self.model_dump(), f, indent=4)
