"""`TrainingConfig`: the reference's train_configs/*.json interface (/root/reference trainer/config.py:38-177) kept
field-for-field (names, defaults, derived values, JSON round trip) so existing config files drive this engine unchanged.

Differences, all deliberate: the device is NOT chosen by probing free memory after torch is initialised
(trainer/utils/utils.py:64-89, racy for parallel launches - SURVEY.md App. C14); the launcher pins one GPU per job through
HIP_VISIBLE_DEVICES (parallel.py) and the job always uses cuda:0.  `pretrained_model` may also name a synthetic
random-init model ("synthetic:sdxl" / "synthetic:sd15" / "synthetic:tinyxl" ...) because this environment has no network.
"""
import json
import os
import time
from datetime import datetime
from typing import List, Literal, Optional, Union

from pydantic import BaseModel

# trainer/config.py:33-36 (paths are resolved by the host application; version is what matters here)
pretrained_models = {
    "sdxl": {"path": "models/checkpoints/juggernautXL_v6.safetensors", "url": None, "version": "sdxl"},
    "sd15": {"path": "models/checkpoints/juggernaut_reborn.safetensors", "url": None, "version": "sd15"},
}


def remove_delimiter_characters(name: str) -> str:
    """trainer/checkpoint.py:58-81: make a run name safe for file names."""
    if name is None:
        return name
    for ch in ["\\", "/", ":", "*", "?", '"', "<", ">", "|", " ", "\n", "\t", "."]:
        name = name.replace(ch, "_")
    return name


class TrainingConfig(BaseModel):
    lora_training_urls: str
    concept_mode: Literal["face", "style", "object"]
    caption_prefix: str = ""
    prompt_modifier: Optional[str] = None
    caption_model: Literal["gpt4-v", "blip", "florence", "no_caption"] = "florence"
    caption_dropout: float = 0.1
    sd_model_version: Optional[Literal["sdxl", "sd15"]] = None
    ckpt_path: Optional[str] = None
    pretrained_model: Optional[dict] = None
    seed: Union[int, None] = None
    resolution: int = 512
    validation_img_size: Optional[Union[int, List[int]]] = None
    train_img_size: Optional[List[int]] = None
    train_aspect_ratio: Optional[float] = None
    train_batch_size: int = 4
    max_train_steps: int = 300
    num_train_epochs: Optional[int] = None
    checkpointing_steps: int = 10000
    gradient_accumulation_steps: int = 1
    is_lora: bool = True

    unet_optimizer_type: Literal["adamw", "prodigy", "AdamW8bit"] = "adamw"
    unet_lr_warmup_steps: Optional[int] = None
    unet_lr: float = 0.0003
    prodigy_d_coef: float = 1.0
    unet_prodigy_growth_factor: float = 1.05
    lora_weight_decay: float = 0.004

    ti_lr: float = 0.001
    token_warmup_steps: int = 0
    ti_weight_decay: float = 0.0
    ti_optimizer: Literal["adamw", "prodigy"] = "adamw"
    freeze_ti_after_completion_f: float = 0.7
    freeze_unet_before_completion_f: float = 0.0

    token_attention_loss_w: float = 3e-7
    cond_reg_w: float = 0.0e-5
    tok_cond_reg_w: float = 0.0e-5
    tok_cov_reg_w: float = 0.0
    l1_penalty: float = 0.03

    noise_offset: float = 0.02
    snr_gamma: Optional[float] = 5.0
    lora_alpha_multiplier: float = 1.0
    lora_rank: int = 16
    use_dora: bool = False

    left_right_flip_augmentation: bool = True
    augment_imgs_up_to_n: int = 40
    mask_target_prompts: Union[None, str] = None
    crop_based_on_salience: bool = True
    use_face_detection_instead: bool = False
    clipseg_temperature: float = 0.5
    n_sample_imgs: int = 4
    name: Optional[str] = None
    output_dir: str = "eden_lora_training_runs"
    debug: bool = False
    allow_tf32: bool = True
    disable_ti: bool = False
    skip_gpt_cleanup: bool = False
    weight_type: Literal["fp16", "bf16", "fp32"] = "bf16"
    n_tokens: int = 3
    inserting_list_tokens: List[str] = ["<s0>", "<s1>", "<s2>"]
    token_dict: dict = {"TOK": "<s0><s1><s2>"}
    device: str = "cuda:0"
    sample_imgs_lora_scale: Optional[float] = None
    dataloader_num_workers: int = 0
    training_attributes: dict = {}
    aspect_ratio_bucketing: bool = False
    start_time: float = 0.0
    job_time: float = 0.0
    text_encoder_lora_optimizer: Union[None, Literal["adamw"]] = None
    text_encoder_lora_lr: float = 1.0e-5
    txt_encoders_lr_warmup_steps: int = 200
    text_encoder_lora_weight_decay: float = 1.0e-5
    text_encoder_lora_rank: int = 16

    model_config = {"extra": "ignore"}      # unknown keys are silently ignored, like the reference's callers rely on

    def __init__(self, **data):
        make_dirs = data.pop("_make_dirs", True)
        super().__init__(**data)
        if not self.ckpt_path:
            self.pretrained_model = pretrained_models.get(self.sd_model_version) if self.pretrained_model is None else self.pretrained_model
        else:
            self.pretrained_model = {"path": self.ckpt_path, "url": None, "version": None}
        if not self.name:
            self.name = os.path.basename(self.lora_training_urls)[:40]
        self.name = remove_delimiter_characters(self.name)
        timestamp = datetime.now().strftime("%d%b_%H%M")
        self.output_dir = self.output_dir + f"/{self.name}_{timestamp}-{self.concept_mode}_res{self.resolution}_{self.max_train_steps}steps"
        if make_dirs:
            os.makedirs(self.output_dir, exist_ok=True)
        if self.seed is None:
            self.seed = int(time.time())
        if self.unet_lr_warmup_steps is None:
            self.unet_lr_warmup_steps = self.max_train_steps
        if self.checkpointing_steps < 1:
            self.checkpointing_steps = self.max_train_steps
        if self.concept_mode == "face":
            self.left_right_flip_augmentation = False
            self.mask_target_prompts = "face"
        if self.use_dora:
            self.l1_penalty = 0.0
            self.lora_weight_decay = 0.0
            self.text_encoder_lora_weight_decay = 0.0
        self.inserting_list_tokens = [f"<s{i}>" for i in range(self.n_tokens)]
        self.token_dict = {"TOK": "".join(self.inserting_list_tokens)}
        self.device = "cuda:0"          # one visible device per job (parallel.job_env)
        self.start_time = time.time()

    @classmethod
    def from_json(cls, file_path: str, **overrides):
        with open(file_path, "r") as f:
            data = json.load(f)
        data.update(overrides)
        return cls(**data)

    def save_as_json(self, file_path: str) -> None:
        with open(file_path, "w") as f:
            json.dump(self.model_dump(), f, indent=4)
