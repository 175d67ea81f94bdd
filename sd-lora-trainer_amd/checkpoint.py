"""Checkpoint writer in the reference's on-disk contract (trainer/checkpoint.py:104-221, 84-102;
trainer/embedding_handler.py:401-422), consumed by stock ComfyUI `LoraLoader` + `embedding:NAME` and A1111:

  {name}_{ver}_lora.safetensors        kohya keys  lora_unet_<module path, '.'->'_'>.lora_down.weight [r,Cin(,3,3)]
                                                   ....lora_up.weight [Cout,r(,1,1)]   ....alpha (scalar = r)
  {name}_{ver}_embeddings.safetensors  clip_l [n,768] (+ clip_g [n,1280] for SDXL)
  special_params.json                  {"TOK": "<s0><s1><s2>"}
  training_args.json                   the TrainingConfig

The peft -> diffusers -> kohya key conversion the reference delegates to diffusers/peft (not installed here) is restated:
peft `base_model.model.<path>.lora_A.weight` -> kohya `lora_unet_<path_>.lora_down.weight`, with the reference's own
`base_model_model_` strip (checkpoint.py:93-100); alpha follows diffusers' `convert_state_dict_to_kohya`
(= number of rows of lora_down, i.e. the rank) [3P-unverified, SURVEY.md 8f-1].
DoRA (use_dora): peft's `lora_magnitude_vector` leaves as `<key>.dora_scale` in the parameter's own shape ([Cout] / [1, Cout, 1, 1]),
the name diffusers' convert_state_dict_to_kohya and the kohya / ComfyUI loaders use for it [3P-unverified, diffusers 0.29.2].
"""
import json
import os

import torch
from safetensors.torch import load_file, save_file

from .config import remove_delimiter_characters


def kohya_key(module_path: str) -> str:
    return "lora_unet_" + module_path.replace(".", "_")


def kohya_text_key(module_path: str) -> str:
    """Text-encoder adapters (checkpoint.py:177-181, 188-199 hand them to diffusers' save_lora_weights as
    text_encoder / text_encoder_2 layers): diffusers' convert_state_dict_to_kohya maps those prefixes to lora_te1_ /
    lora_te2_ [3P-unverified, diffusers 0.29.2]."""
    for pre, k in (("text_encoder_2.", "lora_te2_"), ("text_encoder.", "lora_te1_")):
        if module_path.startswith(pre):
            return k + module_path[len(pre):].replace(".", "_")
    raise ValueError(f"not a text-encoder module: {module_path}")


DTYPES = {"fp16": torch.float16, "bf16": torch.bfloat16, "fp32": torch.float32}


def lora_to_kohya(lora_dict, dtype=torch.bfloat16, key=kohya_key):
    """lora_dict: module -> (A, B) or (A, B, magnitude) in peft layout (LoraArena.export()).  Returns the kohya state dict.
    dtype: the reference saves the adapter tensors as they are trained, i.e. in `config.weight_type` (bf16 by default, config.py:99;
    peft 0.10 creates the adapters in the base layer's dtype); `.alpha` is diffusers' `torch.tensor(len(lora_down))`: a 0-dim int64."""
    sd = {}
    for mod, (A, B, *m) in lora_dict.items():
        k = key(mod)
        sd[k + ".lora_down.weight"] = A.detach().to(dtype).contiguous()
        sd[k + ".lora_up.weight"] = B.detach().to(dtype).contiguous()
        sd[k + ".alpha"] = torch.tensor(int(A.shape[0]))
        if m:
            sd[k + ".dora_scale"] = m[0].detach().to(dtype).contiguous()
    return sd


def kohya_to_lora(sd):
    """Inverse of lora_to_kohya for a given set of module paths is ambiguous ('_' vs '.'); callers pass the targets."""
    out = {}
    for k in sd:
        if k.endswith(".lora_down.weight"):
            base = k[: -len(".lora_down.weight")]
            out[base] = (sd[k].float(), sd[base + ".lora_up.weight"].float()) + ((sd[base + ".dora_scale"].float(),) if base + ".dora_scale" in sd else ())
    return out


def save_checkpoint(output_dir, global_step, arena, ti_rows, token_dict, name, pretrained_model_version, config=None,
                    txt_encoder_keys=("clip_l", "clip_g"), text_arena=None, unet_weights=None):
    """arena: unet.LoraArena (LoRA) ; ti_rows: list of [n_tokens, D] tensors per text encoder (or None);
    text_arena: the text encoders' LoraArena when they are LoRA-trained (same file, lora_te1_/lora_te2_ keys)."""
    os.makedirs(output_dir, exist_ok=True)
    name = remove_delimiter_characters(name)
    files = {}
    # every tensor leaves in the training dtype of the reference (`weight_type`, bf16 unless the config says otherwise): the token rows
    # are rows of the text encoders' own tables there (embedding_handler.py:401-422), the adapters peft modules of a bf16 UNet
    dtype = DTYPES[getattr(config, "weight_type", None) or "bf16"]
    if ti_rows:
        emb = {txt_encoder_keys[i]: r.detach().to(dtype).cpu().contiguous() for i, r in enumerate(ti_rows)}
        files["embeddings"] = os.path.join(output_dir, f"{name}_{pretrained_model_version}_embeddings.safetensors")
        save_file(emb, files["embeddings"])
    with open(os.path.join(output_dir, "special_params.json"), "w") as f:
        json.dump(token_dict, f)
    if arena is not None:
        files["lora"] = os.path.join(output_dir, f"{name}_{pretrained_model_version}_lora.safetensors")
        sd = lora_to_kohya(arena.export(), dtype=dtype)
        if text_arena is not None:
            sd.update(lora_to_kohya(text_arena.export(), dtype=dtype, key=kohya_text_key))
        save_file(sd, files["lora"])
        # adapter_config.json (peft `save_pretrained`, checkpoint.py:175) - the fields the reference's loader reads
        with open(os.path.join(output_dir, "adapter_config.json"), "w") as f:
            json.dump({"peft_type": "LORA", "r": arena.rank, "lora_alpha": arena.rank * arena.scale, "init_lora_weights": "gaussian",
                       "target_modules": ["to_k", "to_q", "to_v", "to_out.0", "conv2"], "use_dora": bool(getattr(arena, "dora", False))}, f, indent=2)
    if unet_weights is not None:
        # is_lora == False (checkpoint.py:210-212): `unet.save_pretrained(output_dir)` = the whole fine-tuned UNet under its
        # diffusers parameter names in diffusion_pytorch_model.safetensors
        files["unet"] = os.path.join(output_dir, "diffusion_pytorch_model.safetensors")
        save_file({k: v.to(dtype).contiguous() for k, v in unet_weights.export().items()}, files["unet"])
        # ... and its config.json (`ModelMixin.save_pretrained` -> `save_config`): what `UNet2DConditionModel.from_pretrained(output_dir)` reads
        from . import topology
        with open(os.path.join(output_dir, "config.json"), "w") as f:
            json.dump(topology.diffusers_unet_config(topology.CONFIGS[pretrained_model_version]), f, indent=2, sort_keys=True)
    if config is not None:
        config.save_as_json(os.path.join(output_dir, "training_args.json"))
    return files


def load_embeddings(path, txt_encoder_keys=("clip_l", "clip_g")):
    sd = load_file(path)
    return [sd[k] for k in txt_encoder_keys if k in sd]


def load_lora(path, targets):
    """-> module -> (A, B) (or (A, B, magnitude) for a DoRA file) for the given module paths (topology.lora_targets)."""
    sd = load_file(path)
    return {m: (sd[kohya_key(m) + ".lora_down.weight"].float(), sd[kohya_key(m) + ".lora_up.weight"].float())
            + ((sd[kohya_key(m) + ".dora_scale"].float(),) if kohya_key(m) + ".dora_scale" in sd else ()) for m in targets}
