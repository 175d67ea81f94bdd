"""MI355X-native LoRA / textual-inversion training-step engine for SD1.5 / SDXL.

Import name: `sd_lora_trainer_amd` (the directory on disk is `sd-lora-trainer_amd/`; the top-level
`sd_lora_trainer_amd/` shim maps the importable name onto it).
"""
from . import topology  # noqa: F401

__all__ = ["topology"]
