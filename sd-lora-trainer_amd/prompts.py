"""Prompt handling around the validation render (host strings only): the reference's trigger-token replacement policy
(/root/reference trainer/inference.py:40-127 `prepare_prompt_for_lora`), the string helpers it uses
(trainer/utils/utils.py:27-47 `replace_in_string`, `fix_prompt`), the with / without-concept prompt pair of
`encode_prompt_advanced` (inference.py:230-286) and the choice of validation prompts of `render_images` (inference.py:312-327).
Pinned against the reference's own functions by tests/golden/prompts.json (oracle/gen_golden2.py).

The reference's validation prompt LISTS (trainer/utils/val_prompts.py) are its data, not reproduced here: `VAL_PROMPTS` below is
a small list of this engine's own in the same format (`<concept>` marks where the learned concept goes); a job can bring any list
through `training_attributes["validation_prompts"]` = {"style": [...], "face": [...], "object": [...]}.
"""
import random
import re

NEGATIVE_PROMPT = ("nude, naked, poorly drawn face, ugly, tiling, out of frame, extra limbs, disfigured, deformed body, blurry, blurred, "
                   "watermark, text, grainy, signature, cut off, draft")     # inference.py:362: the render's fixed negative prompt

VAL_PROMPTS = {
    "style": ["a lighthouse on a cliff above a stormy sea", "a quiet street in an old town after the rain", "a bowl of fruit on a wooden table, morning light",
              "a fox resting in a forest clearing", "a city skyline at dusk seen from a bridge", "a portrait of an old sailor", "a mountain lake under a starry sky",
              "a steam locomotive crossing a viaduct"],
    "face": ["<concept> as a bronze bust in a museum", "a watercolor portrait of <concept>", "<concept> as an astronaut floating above the earth",
             "a black and white film still of <concept> in a train station", "<concept> as a character in a comic book", "a studio photo of <concept> wearing a red scarf",
             "<concept> painted on a brick wall as street art", "a claymation figure of <concept>"],
    "object": ["<concept> on a pedestal in an art gallery", "a watercolor painting of <concept>", "<concept> made of folded paper", "a neon sign in the shape of <concept>",
               "<concept> carved from a block of ice", "a pencil sketch of <concept> in a notebook", "<concept> as a mosaic of small tiles", "a toy version of <concept> on a shelf"],
}


def replace_in_string(s, replacements):
    """utils.py:27-37: case-insensitive regex replacement of every key, repeated until nothing changes."""
    while True:
        replaced = False
        for target, replacement in replacements.items():
            new_s = re.sub(target, replacement, s, flags=re.IGNORECASE)
            if new_s != s:
                s, replaced = new_s, True
        if not replaced:
            return s


def fix_prompt(prompt):
    """utils.py:39-47: whitespace / comma / period clean-up."""
    if not prompt:
        return prompt
    prompt = re.sub(r"\s+", " ", prompt)
    prompt = re.sub(r",,", ",", prompt)
    prompt = re.sub(r"\s?,\s?", ", ", prompt)
    prompt = re.sub(r"\s?\.\s?", ". ", prompt)
    return prompt.strip()


def prepare_prompt_for_lora(prompt, token_map, trigger_text, lora_name, mode, interpolation=False):
    """inference.py:40-127 with the values the reference reads from `special_params.json` (token_map) and
    `training_args.json` (trigger_text, name, concept_mode) passed in."""
    lora_name = str(lora_name) if lora_name is not None else "concept"
    enc = "<" + lora_name + ">"
    if mode != "style":
        prompt = replace_in_string(prompt, {"<concept>": trigger_text, "<concepts>": trigger_text + "'s", enc: trigger_text, enc.lower(): trigger_text,
                                            lora_name: trigger_text, lora_name.lower(): trigger_text})
        if trigger_text not in prompt:
            prompt = trigger_text + ", " + prompt
    else:
        prompt = replace_in_string(prompt, {"in the style of <concept>": "in the style of TOK", f"in the style of {enc}": "in the style of TOK",
                                            f"in the style of {enc.lower()}": "in the style of TOK", f"in the style of {lora_name}": "in the style of TOK",
                                            f"in the style of {lora_name.lower()}": "in the style of TOK"})
        if "in the style of TOK" not in prompt:
            prompt = "in the style of TOK, " + prompt
    prompt = replace_in_string(prompt, {"<concept>": "TOK", enc: "TOK"})
    if interpolation and mode != "style":
        prompt = "TOK, " + prompt
    return fix_prompt(replace_in_string(prompt, token_map))


def prompt_pair(prompt, token_map, trigger_text, lora_name, concept_mode, use_lora=True):
    """encode_prompt_advanced (inference.py:243-257): (prompt WITH the learned tokens, prompt WITHOUT the concept)."""
    lora_prompt = prepare_prompt_for_lora(prompt, token_map, trigger_text, lora_name, concept_mode) if use_lora else prompt
    replace_str = {"face": "person", "object": "object"}.get(concept_mode, "")
    return lora_prompt, fix_prompt(prompt.replace("<concept>", replace_str))


def validation_prompts(concept_mode, n_imgs, seed, prompt_modifier=None, lists=None):
    """render_images (inference.py:309-327): `random.seed(seed); random.sample(list, n)`, the first prompt replaced by "" (style) or
    "<concept>", then `prompt_modifier.format(prompt)`."""
    lists = lists or VAL_PROMPTS
    random.seed(seed)
    key = concept_mode if concept_mode in ("style", "face") else "object"
    raw = random.sample(lists[key], n_imgs)
    raw[0] = "" if key == "style" else "<concept>"
    if prompt_modifier:
        raw = [prompt_modifier.format(p) for p in raw]
    return raw
