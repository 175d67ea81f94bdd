"""`train(config)`: the reference's orchestrator interface (/root/reference main.py:34-551) for the part this engine owns - a
*generator* that yields progress floats and returns `(config, output_save_dir)` through StopIteration.value, driven by the same
train_configs/*.json files:   python -m sd_lora_trainer_amd.train cfg.json

What it does, in the reference's order: load the models (UNet, CLIP text encoder(s), VAE, tokenizer(s): main.py:39-48), add and
initialise the trigger tokens (:92-101), optional token warm-up (:105), the adapters / the full fine-tune and their optimizers
(:116-176), the latent cache - every image VAE-encoded once (:183-191) -, the step loop with its schedules, caption dropout, noise
offset and gradient accumulation (:258-382), checkpoints + validation renders every `checkpointing_steps` (:399-452) and the final
save (:466-533).  Out of scope: preprocessing / captioning / masking - the data source is what `preprocess` leaves
behind: a folder with `captions.csv` (image_path, caption[, mask_path]).

`pretrained_model` (config): {"path": UNet weights (.safetensors, diffusers names) | "synthetic:<version>",
  "text_encoder_path", "text_encoder_2_path": Hugging Face CLIPTextModel / CLIPTextModelWithProjection state dicts,
  "vae_path": AutoencoderKL state dict, "tokenizer_path", "tokenizer_2_path": directories with vocab.json + merges.txt}.
"synthetic:<version>" builds seeded random weights of the exact architecture (there is no network here); a real UNet checkpoint
without text-encoder weights is an error.  `lora_training_urls`: a preprocessed folder, a `.pt` cache of pre-tokenised tensors
(`posterior` | `latents`, `masks`, `input_ids`, `token_lists`), or "synthetic:<n>".
"""
import json
import math
import os
import shutil
import sys
import time

import numpy as np
import torch

from . import checkpoint as ckpt
from . import prompts as P
from . import schedule, topology
from .config import TrainingConfig
from .dataset import DiagonalGaussian
from .embedding_handler import TokenEmbeddingsHandler
from .optimizer import OptimizerCollection


def _random_state(shapes, device, seed, emb_scale=0.02):
    g = torch.Generator(device=device).manual_seed(seed)
    sd = {}
    for n, shp in shapes.items():
        t = torch.randn(shp, generator=g, device=device, dtype=torch.float32)
        if "embedding" in n:
            t *= emb_scale
        elif len(shp) >= 2:
            t *= 1.0 / math.sqrt(math.prod(shp[1:]))
        else:
            t *= 0.02
            if ("norm" in n) and n.endswith(".weight"):
                t += 1.0
        sd[n] = t
    return sd


def _load_state(path):
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file
        return load_file(path)
    return torch.load(path, map_location="cpu")


def _rank_world():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


class Models:
    """What `load_models` returns in the reference (trainer/models.py:7-54), for the parts this engine runs."""

    def __init__(self, config, rt):
        from . import clip as CL
        from . import step as S
        from . import unet as M
        from .tokenizer import ClipBpeTokenizer
        pm = config.pretrained_model or {}
        path = pm.get("path", "")
        self.synthetic = path.startswith("synthetic:")
        self.config, self.rt, self.pm = config, rt, pm
        if self.synthetic:
            self.version = path.split(":", 1)[1]
            self.cfg = topology.CONFIGS[self.version]
        else:
            sd = _load_state(path)
            self.version = config.sd_model_version or ("sdxl" if "add_embedding.linear_1.weight" in sd else "sd15")
            self.cfg = topology.CONFIGS[self.version]
        cfg, n_tok = self.cfg, config.n_tokens
        xl, tiny = bool(cfg["addition"]), self.version.startswith("tiny")
        self.kinds = (["tiny_l", "tiny_g"] if tiny else ["clip_l", "clip_g"]) if xl else (["tiny_l"] if tiny else ["clip_l"])
        # ---- tokenizers (trainer/models.py: pipe.tokenizer / tokenizer_2) + the trigger tokens (embedding_handler.py:176-180)
        self.tokenizers = None
        if pm.get("tokenizer_path"):
            t1 = ClipBpeTokenizer.from_files(pm["tokenizer_path"])
            toks = [t1]
            if xl:      # SDXL's tokenizer_2 is the same vocabulary padded with "!" (id 0) instead of <|endoftext|>
                toks.append(ClipBpeTokenizer.from_files(pm["tokenizer_2_path"]) if pm.get("tokenizer_2_path")
                            else ClipBpeTokenizer.from_files(pm["tokenizer_path"], **({"pad_token": "!"} if "!" in t1.vocab else {})))
            for t in toks:
                t.add_tokens(config.inserting_list_tokens)
            self.tokenizers = toks
        # ---- UNet
        if self.synthetic:
            sd = _random_state(topology.param_shapes(cfg), rt.device, seed=config.seed)
        if config.is_lora:
            self.unet = M.UNet(rt, cfg, sd, lora_rank=config.lora_rank, lora_alpha_multiplier=config.lora_alpha_multiplier, use_dora=config.use_dora)
        else:                          # main.py:144-149: full fine-tune, every UNet parameter trained
            from . import fullft
            self.unet = M.UNet(rt, cfg, sd, trainer=fullft.WeightTrainer(rt))
        del sd
        # ---- text encoders: always built (the reference encodes the captions through them even with disable_ti, main.py:306-308)
        te_arena = None
        if config.text_encoder_lora_optimizer is not None:      # built independently of disable_ti, like main.py:116-131
            te_arena = M.LoraArena(rt, config.text_encoder_lora_rank, config.lora_alpha_multiplier, problems=[], dora=config.use_dora)      # a21
        self.encoders = []
        for i, kd in enumerate(self.kinds):
            c = topology.CLIP_CONFIGS[kd]
            csd = self.clip_state(i)
            self.encoders.append(CL.ClipTextEncoder(rt, f"te{i + 1}", csd, heads=c["heads"], act=c["act"], mode="penultimate" if xl else "last",
                                                    with_projection=bool(c["proj"]), n_train=n_tok, arena=te_arena,
                                                    lora_prefix="text_encoder." if i == 0 else "text_encoder_2."))
        if te_arena is not None:
            te_arena.finalize()
        self.text = S.TextStack(rt, self.encoders, pool_mode="argmax", arena=te_arena)

    def clip_state(self, i):
        """Hugging Face state dict of text encoder i with the n new token rows appended (`resize_token_embeddings`,
        embedding_handler.py:183)."""
        config, pm, kd = self.config, self.pm, self.kinds[i]
        c = topology.CLIP_CONFIGS[kd]
        n_tok = config.n_tokens
        key = "text_encoder_path" if i == 0 else "text_encoder_2_path"
        if self.synthetic and not pm.get(key):
            vocab = (len(self.tokenizers[i]) - n_tok) if self.tokenizers else c["vocab"]
            return _random_state(topology.clip_param_shapes(dict(c, vocab=vocab), n_tok), self.rt.device, seed=config.seed + 1 + i)
        if not pm.get(key):
            raise ValueError(f"pretrained_model['{key}'] is missing: a real UNet checkpoint needs the weights of its text encoder(s) "
                             "(Hugging Face CLIPTextModel state dict); random text encoders are only built for 'synthetic:' models")
        csd = dict(_load_state(pm[key]))
        tk = next(k for k in csd if k.endswith("embeddings.token_embedding.weight"))
        tab = csd[tk].float()
        # transformers initialises the rows `resize_token_embeddings` adds from N(0, initializer_range = 0.02) [3P, version dependent];
        # they are overwritten by initialize_new_tokens, but enter its std target with weight n / V first
        g = torch.Generator().manual_seed(config.seed + 1 + i)
        csd[tk] = torch.cat([tab, 0.02 * torch.randn(n_tok, tab.shape[1], generator=g)], 0)
        return csd

    def vae_state(self):
        pm = self.pm
        if pm.get("vae_path"):
            return _load_state(pm["vae_path"])
        if self.synthetic:
            c = topology.VAE_CONFIGS["tiny" if self.version.startswith("tiny") else "sd"]
            return _random_state(topology.vae_param_shapes(c), self.rt.device, seed=self.config.seed + 77)
        return None

    def unet_state(self):
        """The frozen UNet weights again (for the inference instance of the validation render)."""
        if self.synthetic:
            return _random_state(topology.param_shapes(self.cfg), self.rt.device, seed=self.config.seed)
        return _load_state(self.pm["path"])


def build_models(config, rt):
    """-> (unet, text_stack, version).  Mirrors load_models (trainer/models.py:7-54) for the parts on the step."""
    m = Models(config, rt)
    return m.unet, m.text, m.version


def synthetic_cache(cfg, n_images, h, w, vocab, n_tokens, seed):
    g = torch.Generator().manual_seed(seed)
    tok = list(range(vocab - n_tokens, vocab))
    bos, eos = (49406, 49407) if vocab > 49407 else (vocab - n_tokens - 2, vocab - n_tokens - 1)
    ids = torch.full((n_images, 77), eos, dtype=torch.int64)
    lists = []
    for i in range(n_images):
        words = torch.randint(1, bos - 1, (6,), generator=g).tolist()
        l = [bos] + words[:3] + tok + words[3:] + [eos]
        ids[i, :len(l)] = torch.tensor(l)
        lists.append(l)
    # like the reference's dataset the cache holds the VAE POSTERIOR of each image (mean | logvar), sampled anew on every fetch
    post = torch.cat([torch.randn(n_images, 4, h, w, generator=g), torch.full((n_images, 4, h, w), -4.0)], dim=1)
    return dict(posterior=post,
                masks=(torch.rand(n_images, 1, h, w, generator=g) * 0.95 + 0.05).repeat(1, 4, 1, 1), input_ids=ids, token_lists=lists,
                tok_list=[bos] + tok + [eos], description_ids=[bos] + torch.randint(1, bos - 1, (8,), generator=g).tolist() + [eos])


def _tokenize(tokenizers, texts):
    """-> (ids per tokenizer: int64 [N, 77], token lists of tokenizer 0 as `pipe.tokenizer.encode` gives them, loss.py:32)."""
    ids = [torch.tensor(t(texts), dtype=torch.int64) for t in tokenizers]
    return ids, [tokenizers[0].encode(s) for s in texts]


REG_CAPTIONS = ["a photo of TOK", "TOK", "a photo of TOK next to TOK", "TOK and TOK"]      # ConditioningRegularizer.reg_captions (loss.py:183)


def reg_caption_ids(config, models, cache):
    """Token ids [4, 77] per tokenizer of the tok_cond_reg_w captions with TOK replaced by the trigger string (loss.py:184, 242).
    Without tokenizer files (synthetic jobs) the filler words are fixed ids of the synthetic vocabulary."""
    if models.tokenizers is not None:
        rep = config.token_dict.get("TOK", "TOK")
        return _tokenize(models.tokenizers, [c.replace("TOK", rep) for c in REG_CAPTIONS])[0]
    tl = cache["tok_list"]
    bos, tok, eos = tl[0], tl[1:-1], tl[-1]
    word = lambda i: 1 + (i * 37) % (min(bos, eos) - 1)  # noqa: E731
    caps = [[word(1), word(2), word(3)] + tok, tok, [word(1), word(2), word(3)] + tok + [word(4), word(5)] + tok, tok + [word(6)] + tok]
    ids = torch.full((len(caps), 77), eos, dtype=torch.int64)
    for r, c in enumerate(caps):
        ids[r, 0] = bos
        ids[r, 1:1 + len(c)] = torch.tensor(c)
    return [ids] * len(models.encoders)


def load_data(config, models, rt, h, w):
    """The job's latent / mask / caption cache -> dict(posterior | latents, masks, input_ids (list per tokenizer), token_lists,
    tok_ids (list per tokenizer), tok_list, description_ids?, captions?)."""
    src = config.lora_training_urls
    cfg, n_enc = models.cfg, len(models.encoders)
    if src.startswith("synthetic:"):
        vocab = models.encoders[0].V
        cache = synthetic_cache(cfg, int(src.split(":")[1] or 8), h, w, vocab, config.n_tokens, config.seed)
    elif os.path.isdir(src):
        # PreprocessedDataset (dataset.py:31-90; main.py:183-191): every image through the VAE encoder ONCE, the posterior kept
        from . import unet as M
        from . import vae as V
        from .dataset import LatentCache
        if models.tokenizers is None:
            raise ValueError("an image folder needs the tokenizer files: pretrained_model['tokenizer_path'] (vocab.json + merges.txt)")
        vsd = models.vae_state()
        if vsd is None:
            raise ValueError("an image folder needs the VAE weights: pretrained_model['vae_path']")
        enc = V.VaeEncoder(M.Runtime(rt.device, 1, act_dtype=rt.act, ops=rt.ops), vsd)
        f = 2 ** (len(enc.downs) - 1)          # 8 for the SD / SDXL VAE (4 levels); the toy VAE of the tests has 3
        lc = LatentCache.from_folder(src, enc, size=(f * w, f * h), scaling_factor=cfg["scaling_factor"], substitute_caption_map=config.token_dict)
        del enc, vsd
        if rt.device.type == "cuda":
            torch.cuda.empty_cache()
        ids, lists = _tokenize(models.tokenizers, lc.captions)
        cache = dict(posterior=torch.cat([d.parameters for d in lc.dists], 0), masks=torch.stack(lc.masks), input_ids=ids, token_lists=lists, captions=lc.captions)
    else:
        cache = torch.load(src)
    if not isinstance(cache["input_ids"], (list, tuple)):
        cache["input_ids"] = [cache["input_ids"]] * n_enc
    if models.tokenizers is not None:
        tok_ids, tok_lists = _tokenize(models.tokenizers, [config.token_dict["TOK"]])            # the caption-dropout caption (main.py:304)
        cache["tok_ids"], cache["tok_list"] = [t[0] for t in tok_ids], tok_lists[0]
        desc = (config.training_attributes or {}).get("gpt_description")
        if desc:
            cache["description_ids"] = [t[0] for t in _tokenize(models.tokenizers, [desc])[0]]
    elif cache.get("tok_list") is not None:
        eos = int(cache["input_ids"][0][0, -1])
        row = torch.full((77,), eos, dtype=torch.int64)
        row[:len(cache["tok_list"])] = torch.tensor(cache["tok_list"])
        cache["tok_ids"] = [row] * n_enc
        if cache.get("description_ids") is not None and not torch.is_tensor(cache["description_ids"][0]):
            d = torch.full((77,), eos, dtype=torch.int64)
            d[:len(cache["description_ids"])] = torch.tensor(cache["description_ids"])
            cache["description_ids"] = [d] * n_enc
    return cache


class Renderer:
    """`render_images` (trainer/inference.py:289-406) on this engine: validation prompts -> with / without-concept conditionings
    (encode_prompt_advanced + blend_conditions) -> Euler-trailing CFG sampler on an inference instance of the UNet (batch 2) with the
    current adapters -> VAE decode -> JPEGs + validation grid.  Built lazily at the first checkpoint."""

    def __init__(self, config, models, train_unet):
        from . import clip as CL
        from . import sampler as SM
        from . import step as S
        from . import unet as M
        from . import vae as V
        self.config, self.models, self.train_unet = config, models, train_unet
        cfg = models.cfg
        dev = models.rt.device
        self.rt = M.Runtime(dev, 2)
        self.unet = M.UNet(self.rt, cfg, models.unet_state(), lora_rank=config.lora_rank, lora_alpha_multiplier=config.lora_alpha_multiplier,
                           use_dora=config.use_dora)
        self.sampler = SM.LatentSampler(self.rt, self.unet)
        self.decoder = V.VaeDecoder(M.Runtime(dev, 1), models.vae_state())
        xl = bool(cfg["addition"])
        self.encoders = []
        # text-encoder LoRA (text_encoder_lora_optimizer): the reference renders with the pipe's own peft-wrapped text encoders
        # (inference.py:345-356), so the inference encoders carry the same adapters, refreshed from the training arena in sync()
        self.te_arena = None
        if models.text.arena is not None:
            self.te_arena = M.LoraArena(self.rt, config.text_encoder_lora_rank, config.lora_alpha_multiplier, problems=[], dora=config.use_dora)
        for i, kd in enumerate(models.kinds):
            c = topology.CLIP_CONFIGS[kd]
            self.encoders.append(CL.ClipTextEncoder(self.rt, f"rte{i + 1}", models.clip_state(i), heads=c["heads"], act=c["act"],
                                                    mode="penultimate" if xl else "last", with_projection=bool(c["proj"]), n_train=config.n_tokens,
                                                    arena=self.te_arena, lora_prefix="text_encoder." if i == 0 else "text_encoder_2."))
        if self.te_arena is not None:
            self.te_arena.finalize()
        self.text = S.TextStack(self.rt, self.encoders, pool_mode="argmax", arena=self.te_arena)
        self.ctx = self.rt.zeros(2 * M.CTX_PAD, cfg["cross_dim"])

    def sync(self):
        """Current adapters and token rows of the training instance -> the inference instance."""
        self.unet.arena.params.copy_(self.train_unet.arena.params)
        self.unet.arena.refresh_shadows()
        if self.te_arena is not None:
            self.te_arena.params.copy_(self.models.text.arena.params)
            self.te_arena.refresh_shadows()
        n = self.config.n_tokens
        for dst, src in zip(self.encoders, self.models.encoders):
            dst.table[dst.V - n:].copy_(src.table[src.V - n:])

    def encode(self, prompt, negative):
        """pipe.encode_prompt(prompt, do_classifier_free_guidance=True, negative_prompt=...) -> (c, uc[, pc, puc]); batch row 0 = negative."""
        from .unet import CTX_PAD
        ids = [torch.tensor(t([negative, prompt]), dtype=torch.int64) for t in self.models.tokenizers]
        self.text.set_ids([i.to(self.rt.device) for i in ids])
        pooled = self.text.forward(self.ctx)
        cv = self.ctx.view(2, CTX_PAD, -1)[:, :77].float().clone()
        out = (cv[1:2], cv[0:1])
        if pooled is not None:
            pf = pooled.float().clone()
            out += (pf[1:2], pf[0:1])
        return out

    @torch.no_grad()
    def render(self, out_dir, train_step, n_steps=25):
        from . import sampler as SM
        config, cfg = self.config, self.models.cfg
        self.sync()
        lists = (config.training_attributes or {}).get("validation_prompts")
        raw = P.validation_prompts(config.concept_mode, config.n_sample_imgs, config.seed, config.prompt_modifier, lists if isinstance(lists, dict) else None)
        trig = (config.training_attributes or {}).get("trigger_text", "TOK")
        embeds, used = [], []
        for p in raw:
            lora_p, zero_p = P.prompt_pair(p, config.token_dict, trig, config.name, config.concept_mode, use_lora=not config.disable_ti)
            # render_images passes token_scale = 0 with disable_ti (inference.py:289-385): the conditioning is the zero prompt's alone
            e, _ = SM.blend_conditions(self.encode(zero_p, P.NEGATIVE_PROMPT), self.encode(lora_p, P.NEGATIVE_PROMPT), config.sample_imgs_lora_scale,
                                       token_scale=0.0 if config.disable_ti else None)
            embeds.append(e)
            used.append(lora_p)
        size = config.validation_img_size
        size = (size, size) if isinstance(size, int) else tuple(size)
        paths = SM.render_images(self.sampler, self.decoder, embeds, size, out_dir, train_step, config.seed, scaling_factor=cfg["scaling_factor"],
                                 lora_scale=config.sample_imgs_lora_scale, n_steps=n_steps)
        make_validation_img_grid(paths, os.path.join(out_dir, "validation_grid.jpg"))
        return raw


def make_validation_img_grid(paths, out_path):
    """trainer/utils/io.py:99-136: the renders of one checkpoint side by side (2 columns)."""
    from PIL import Image
    if not paths:
        return None
    imgs = [Image.open(p) for p in paths]
    w, h = imgs[0].size
    cols = 2 if len(imgs) > 1 else 1
    rows = (len(imgs) + cols - 1) // cols
    grid = Image.new("RGB", (cols * w, rows * h))
    for i, im in enumerate(imgs):
        grid.paste(im, ((i % cols) * w, (i // cols) * h))
    grid.save(out_path, format="JPEG", quality=95)
    return out_path


def train(config: TrainingConfig, runtime=None, every_step=False):
    """Generator: yields progress in [0,1]; returns (config, output_save_dir).  every_step: also yield (None) after every
    optimizer call, so that several jobs can be advanced in lock-step by train_concurrent."""
    from . import step as S
    from . import unet as M
    if config.aspect_ratio_bucketing:
        raise NotImplementedError("aspect_ratio_bucketing (broken in the reference as well, README.md:76) is not built in this engine: the step's "
                                  "graphs are captured for one latent shape; refusing to train something else silently")
    rank, world = _rank_world()
    ddp = (not config.is_lora) and world > 1
    if ddp and str(config.device).startswith("cuda") and os.environ.get("LOCAL_RANK") is not None and torch.cuda.device_count() > 1:
        config.device = f"cuda:{int(os.environ['LOCAL_RANK'])}"         # one process per GPU; a launcher that pins HIP_VISIBLE_DEVICES leaves cuda:0
    np.random.seed(config.seed + (rank if ddp else 0))                  # caption dropout: each data-parallel rank draws its own
    torch.manual_seed(config.seed)
    B = config.train_batch_size
    rt = runtime or M.Runtime(config.device, B)
    models = Models(config, rt)
    unet, version, cfg = models.unet, models.version, models.cfg
    config.sd_model_version = version if version in ("sdxl", "sd15") else config.sd_model_version
    config.pretrained_model = dict(config.pretrained_model or {}, version=version)
    if not config.sample_imgs_lora_scale:                                # main.py:57-67
        config.sample_imgs_lora_scale = 0.75 if cfg["addition"] else 0.85
    if not config.validation_img_size:
        config.validation_img_size = 1024 if cfg["addition"] else 768
    if config.train_img_size is None:
        config.train_img_size = [config.resolution, config.resolution]
    w, h = config.train_img_size[0] // 8, config.train_img_size[1] // 8
    cache = load_data(config, models, rt, h, w)
    n_img = (cache["posterior"] if "posterior" in cache else cache["latents"]).shape[0]
    # DataLoader(drop_last=False): main.py:200-207.  Data parallel: every rank takes ceil(n_img / world) samples of the shared shuffle
    # (DistributedSampler semantics: the permutation wraps around to a multiple of world), so ALL ranks run the same number of steps -
    # every step issues collectives and the checkpoint a barrier - and max_train_steps counts optimizer steps, not samples
    shard = math.ceil(n_img / world) if ddp else n_img
    steps_per_epoch = math.ceil(shard / B)
    config.num_train_epochs = math.ceil(config.max_train_steps / steps_per_epoch)

    ti_on = not config.disable_ti
    # The text encoders are part of the step when the token rows train OR when they carry adapters (main.py:116-131 builds the text-encoder LoRA whether or not
    # disable_ti is set).  disable_ti with adapters: the step of a TI run whose token rows never move - no TI optimizer (main.py:133), no token-attention loss
    # (main.py:342) and none of the token regularisers (main.py:357: they need a TI optimizer with lr > 0); the text backward runs for the adapters' gradients.
    text_in_step = ti_on or config.text_encoder_lora_optimizer is not None
    ts = S.TrainStep(rt, unet, latent_hw=(h, w), process_group=True if ddp else None, snr_gamma=config.snr_gamma, l1_penalty=config.l1_penalty, weight_decay=config.lora_weight_decay,
                     grad_accum=config.gradient_accumulation_steps, text=models.text if text_in_step else None, n_tokens=config.n_tokens,
                     token_attention_loss_w=config.token_attention_loss_w if ti_on else 0.0, ti_weight_decay=config.ti_weight_decay if ti_on else 0.0,
                     optimizer=config.unet_optimizer_type, ti_optimizer=config.ti_optimizer if ti_on else "adamw",
                     prodigy_d_coef=config.prodigy_d_coef, prodigy_growth_rate=config.unet_prodigy_growth_factor,
                     text_lora_weight_decay=config.text_encoder_lora_weight_decay,
                     cond_reg_w=config.cond_reg_w if ti_on else 0.0, tok_cov_reg_w=config.tok_cov_reg_w if ti_on else 0.0,
                     tok_cond_reg_w=config.tok_cond_reg_w if ti_on else 0.0, ti_trainable=ti_on,
                     reg_caption_ids=reg_caption_ids(config, models, cache) if (ti_on and config.tok_cond_reg_w > 0.0) else None)
    # main.py:92-101: the new token rows are initialised whether or not they are trained (with disable_ti they stay as drawn)
    from .ti import TiState
    ti_state = ts.ti if text_in_step else TiState(rt, models.encoders, config.n_tokens)
    handler = TokenEmbeddingsHandler(ti_state, config.inserting_list_tokens)
    handler.initialize_new_tokens(seed=config.seed)
    # a20 token warm-up (main.py -> embedding_handler.pre_optimize_token_embeddings, :321-399): only with token_warmup_steps > 0
    # and a description of the concept (training_attributes["gpt_description"], tokenised by the data stage: `description_ids`)
    if ti_on and config.token_warmup_steps > 0 and (config.training_attributes or {}).get("gpt_description") \
            and cache.get("description_ids") is not None and cache.get("tok_ids") is not None:
        warm = ts.token_warmup(cache["tok_ids"], cache["description_ids"], config.token_warmup_steps, config.ti_lr)
        config.training_attributes = dict(config.training_attributes, token_warmup_losses=[warm[0], warm[-1]])
    g = torch.Generator(device=rt.device).manual_seed(config.seed)       # weight / adapter initialisation: shared by data-parallel replicas
    arena = unet.arena
    if arena is not None:
        for e in arena.entries:        # peft init_lora_weights="gaussian" (optimizer.py:89): A ~ N(0, 1/r), B = 0
            e["A"].copy_(torch.randn(e["A"].shape, generator=g, device=rt.device) / config.lora_rank)
            e["B"].zero_()
        arena.refresh_shadows()
    if ts.te_arena is not None:
        for e in ts.te_arena.entries:
            e["A"].copy_(torch.randn(e["A"].shape, generator=g, device=rt.device) / config.text_encoder_lora_rank)
            e["B"].zero_()
        ts.te_arena.refresh_shadows()
    gd = torch.Generator(device=rt.device).manual_seed(config.seed + 1000003 * (rank if ddp else 0))    # data: posterior samples, noise, timesteps
    optimizers = OptimizerCollection(ts, config)
    checkpoint_dir = os.path.join(config.output_dir, "checkpoints")
    if rank == 0:
        if os.path.exists(checkpoint_dir):                               # main.py:222-225
            shutil.rmtree(checkpoint_dir)
        os.makedirs(checkpoint_dir, exist_ok=True)
    time_ids = torch.tensor([[1024., 1024, 0, 0, float(config.resolution), float(config.resolution)]] * B) if cfg["addition"] else None

    # ---- frozen text encoders (disable_ti): the conditioning of every caption is a constant of the job - encode each caption
    # (and the caption-dropout caption) ONCE instead of once per step (the reference re-encodes, main.py:306-308; same values)
    cond = None

    def encode_rows(ids_per_enc):
        n = ids_per_enc[0].shape[0]
        ctxs, pools = [], []
        for s in range(0, n, B):
            chunk = [torch.cat([i[s:s + B], i[-1:].expand(B - min(B, n - s), 77)]) if n - s < B else i[s:s + B] for i in ids_per_enc]
            models.text.set_ids([c.to(rt.device) for c in chunk])
            pooled = models.text.forward(ts.ctx)
            ctxs.append(ts.ctx.view(B, M.CTX_PAD, -1)[:, :77].clone())
            if pooled is not None:
                pools.append(pooled.clone())
        return torch.cat(ctxs)[:n], (torch.cat(pools)[:n] if pools else None)
    if not text_in_step:
        with torch.no_grad():
            cond = encode_rows(cache["input_ids"])
            cond_tok = encode_rows([t.view(1, 77) for t in cache["tok_ids"]]) if cache.get("tok_ids") is not None else None

    renderer = None
    can_render = (rank == 0 and config.n_sample_imgs > 0 and config.is_lora and models.tokenizers is not None and rt.device.type == "cuda"
                  and (models.synthetic or (config.pretrained_model or {}).get("vae_path")))

    def save(step_no, n_steps):
        """save_checkpoint + render_images of one checkpoint (main.py:402-447 / 491-533); rank 0 writes, everyone waits."""
        nonlocal renderer
        out_dir = os.path.join(checkpoint_dir, f"checkpoint-{step_no}")
        vp = None
        if rank == 0:
            ckpt.save_checkpoint(out_dir, step_no, arena, ti_state.rows, config.token_dict, config.name, version, config=config,
                                 text_arena=ts.te_arena, unet_weights=unet.trainer)
            if can_render:
                if renderer is None:
                    renderer = Renderer(config, models, unet)
                vp = renderer.render(out_dir, step_no, n_steps=n_steps)
                shutil.copy(os.path.join(out_dir, "validation_grid.jpg"), os.path.join(checkpoint_dir, f"validation_grid_{step_no:04d}.jpg"))
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        return out_dir, vp

    # ---- the job's data lives on the device: per step only device-side gathers, no host -> device copies and no host syncs, so the
    # host prepares step i+1 while the GPU runs step i (round 1 copied pageable tensors every step: the loop ran 15 % below the replay rate)
    dev = rt.device
    data = cache["posterior"] if "posterior" in cache else cache["latents"]
    data_d, masks_d = data.to(dev), cache["masks"].to(dev)
    has_tok = cache.get("tok_ids") is not None
    N_TOK_ROW = n_img                                                    # row n_img of the per-caption tables = the caption-dropout caption
    ids_tab = lists_all = cap_table = None
    if text_in_step:
        ids_tab = [torch.cat([t, (tok.view(1, 77) if has_tok else t[:1])]).to(dev) for t, tok in zip(cache["input_ids"], cache["tok_ids"] if has_tok else cache["input_ids"])]
        lists_all = list(cache["token_lists"]) + [list(cache["tok_list"]) if has_tok else list(cache["token_lists"][0])]
        cap_table = ts.ta.caption_table(lists_all, models.encoders[0].train_ids.tolist())
    if time_ids is not None:
        time_ids = time_ids.to(dev)
    post_mean = post_std = pool_tab = None
    if "posterior" in cache:
        # dataset.DiagonalGaussian's clamp / exp on the whole table once instead of on the batch's rows every step (same values)
        g_all = DiagonalGaussian(data_d)
        post_mean, post_std = g_all.mean.contiguous(), g_all.std.contiguous()
    if text_in_step:
        pool_tab = ts.text.pool_position_table(ids_tab[-1])

    def cond_rows(cnd, cnd_tok):
        """[n_img + 1] conditioning rows (dataset captions + the dropout caption) on the device."""
        ctx_t = torch.cat([cnd[0], cnd_tok[0][:1] if cnd_tok is not None else cnd[0][:1]]).to(dev)
        pool_t = torch.cat([cnd[1], cnd_tok[1][:1] if cnd_tok is not None else cnd[1][:1]]).to(dev) if cnd[1] is not None else None
        return ctx_t, pool_t
    cond_d = cond_rows(cond, cond_tok) if cond is not None else None

    losses = {"img_loss": [], "tot_loss": []}
    global_step, last_save_step, images_done = 0, 0, 0
    start, pause = time.time(), 0.0
    captured = False
    perm_rng = np.random.RandomState(config.seed)                        # the shuffle is shared; ranks take disjoint slices of it
    done = False
    validation_prompts = None
    for epoch in range(config.num_train_epochs):
        order = perm_rng.permutation(n_img)
        if ddp:
            order = np.resize(order, shard * world)[rank::world]
        spe = steps_per_epoch
        # the epoch's batches: the DataLoader's short last batch wraps around to the start of the epoch's order (fixed-shape step)
        padded = np.resize(order, spe * B) if len(order) < spe * B else order[: spe * B]
        order_d = torch.as_tensor(padded).to(dev).view(spe, B)
        # caption dropout (main.py:300-304): np.random.rand() per sample in loop order - drawn for the whole epoch at once (same stream)
        drop = (np.random.rand(spe, B) < config.caption_dropout) if (config.caption_dropout > 0.0 and has_tok) else np.zeros((spe, B), dtype=bool)
        sel_d = torch.where(torch.as_tensor(drop).to(dev), torch.full((spe, B), N_TOK_ROW, device=dev, dtype=order_d.dtype), order_d)
        for step_in_epoch in range(spe):
            completion_f = schedule.completion_fraction(epoch, step_in_epoch, spe, config.num_train_epochs)
            lrs = schedule.learning_rates(config, global_step, completion_f, ti_active=ti_on, text_lora_active=ts.te_arena is not None)
            if ts.te_arena is not None:
                optimizers.optimizers["text_encoders"].param_groups[0]["lr"] = lrs["text_encoders"]
            optimizers.optimizers["unet"].param_groups[0]["lr"] = lrs["unet"]
            if text_in_step:      # (token rows frozen from the first step with disable_ti)
                optimizers.optimizers["textual_inversion"].param_groups[0]["lr"] = lrs["textual_inversion"] if ti_on else 0.0
            idx, sel = order_d[step_in_epoch], sel_d[step_in_epoch]
            # the batch is written straight into the step's buffers (set_batch skips `x is self.x`): ~25 small launches between two graph
            # replays instead of 53 (tools/train_loop_gaps.py); same draws from the generator in the same order, same arithmetic
            mask = torch.index_select(masks_d, 0, idx, out=ts.mask) if masks_d.dtype == ts.mask.dtype and masks_d.shape[1:] == ts.mask.shape[1:] else masks_d[idx]
            if "posterior" in cache:       # dataset.py:184-187: latent_dist.sample() * scaling_factor on EVERY fetch
                eps = torch.randn(ts.latent.shape, generator=gd, device=dev, dtype=post_mean.dtype)
                smp = post_std[idx].mul_(eps).add_(post_mean[idx])                      # mean + std * noise (separately rounded, as DiagonalGaussian.sample)
                latent = torch.mul(smp, cfg["scaling_factor"], out=ts.latent)
            else:
                latent = torch.index_select(data_d, 0, idx, out=ts.latent) if data_d.dtype == ts.latent.dtype and data_d.shape[1:] == ts.latent.shape[1:] else data_d[idx]
            noise = torch.randn(ts.noise.shape, generator=gd, device=dev, out=ts.noise)
            if config.noise_offset > 0.0:                                                # main.py:313-317
                noise += config.noise_offset * torch.randn((B, 4, 1, 1), generator=gd, device=dev)
            timesteps = torch.randint(0, 1000, (B,), generator=gd, device=dev, out=ts.timesteps)
            if text_in_step:
                kw = {}
                if ti_on and lrs["textual_inversion"] == 0.0 and ts.te_arena is None and ts.prodigy_ti is None and (captured or dev.type != "cuda") and ts._acc is None \
                        and completion_f > config.freeze_ti_after_completion_f:
                    # f4: the token rows are frozen for the rest of the run (main.py:273-274) -> every caption's conditioning is a
                    # constant; encode each caption (and the dropout caption) once with the final rows, then skip the text encoders
                    if cond_d is None:
                        with torch.no_grad():
                            cond = encode_rows(cache["input_ids"])
                            cond_tok = encode_rows([t.view(1, 77) for t in cache["tok_ids"]]) if has_tok else None
                        cond_d = cond_rows(cond, cond_tok)
                    kw = dict(ctx=cond_d[0][sel], pooled=cond_d[1][sel] if cond_d[1] is not None else None)
                ts.set_batch(latent, noise, timesteps, mask, time_ids=time_ids, ids=(ids_tab, sel, pool_tab), caption_table=(cap_table, sel), **kw)
            else:
                ts.set_batch(latent, noise, timesteps, mask, cond_d[0][sel], cond_d[1][sel] if cond_d[1] is not None else None, time_ids)
            if not captured and rt.device.type == "cuda":
                t0 = time.time()
                ts.capture(warmup=1)
                torch.cuda.synchronize()
                pause += time.time() - t0                        # one-off graph capture: not part of the step loop's rate
                captured = True
            optimizers.step(last_batch=step_in_epoch + 1 == spe)
            optimizers.zero_grad()
            if global_step % max(config.max_train_steps // 20, 1) == 0:
                losses["img_loss"].append(float(ts.loss))
                losses["tot_loss"].append(ts.total_loss())
            # main.py:399-452 (fires at step 0 too, App. C3)
            if global_step % config.checkpointing_steps == 0 and global_step < config.max_train_steps - 25:
                t0 = time.time()
                _, vp = save(global_step, 25)
                validation_prompts = vp or validation_prompts
                last_save_step = global_step
                pause += time.time() - t0                        # images_per_second is the step loop's rate, not the renders'
            images_done += B
            global_step += 1
            every = max(config.max_train_steps // 100, 1)         # main.py:457 divides by zero for max_train_steps < 100
            if global_step % every == 0:
                yield float(min(global_step / config.max_train_steps + 0.05, 1.0))
            elif every_step:
                yield None
            if global_step > config.max_train_steps:               # main.py:462 (runs max_train_steps + 1 steps)
                done = True
                break
        if done:
            break
    if rt.device.type == "cuda":
        torch.cuda.synchronize()
    loop_time = time.time() - start - pause
    # final save (main.py:466-533): a fresh checkpoint unless one was written within the last 26 steps
    final_step = global_step if (global_step - last_save_step) > 26 else last_save_step
    output_save_dir = os.path.join(checkpoint_dir, f"checkpoint-{final_step}")
    config.job_time = time.time() - config.start_time
    config.training_attributes = dict(config.training_attributes, images_per_second=images_done / max(loop_time, 1e-9), losses=losses)
    if not os.path.exists(output_save_dir) or world > 1:
        _, vp = save(final_step, 30)
        validation_prompts = vp or validation_prompts
    config.job_time = time.time() - config.start_time
    if validation_prompts is not None:
        config.training_attributes = dict(config.training_attributes, validation_prompts=validation_prompts)
    if rank == 0:
        config.save_as_json(os.path.join(output_save_dir, "training_args.json"))
    return config, output_save_dir


def train_concurrent(configs, on_progress=None, runtimes=None):
    """Several independent jobs on ONE GPU, in one process: every job gets its own stream, weights, adapters and hipGraph, and the
    jobs are advanced round-robin one optimizer call at a time.  At batch 1 the step is bound by per-kernel latency and partial
    waves of workgroups, so the replays of two jobs overlap on the device (+30 % aggregate images/s for two SDXL jobs; two
    PROCESSES on one GPU time-slice instead).  The jobs share numpy's / torch's global RNG (caption dropout), like jobs started
    from one shell script share nothing but the device.  Returns [(config, output_save_dir)] in the order of `configs`."""
    use_cuda = torch.cuda.is_available() and all(str(c.device).startswith("cuda") for c in configs)
    hinted = use_cuda and len(configs) > 1
    if hinted:
        from . import ops as _ops
        _ops.set_throughput_hint(True)       # the GEMM tile heuristics know that other jobs fill the CUs a launch leaves idle
    streams = [torch.cuda.Stream() if use_cuda else None for _ in configs]
    gens = [train(c, runtime=runtimes[i] if runtimes else None, every_step=True) for i, c in enumerate(configs)]
    results, live = [None] * len(configs), set(range(len(configs)))
    try:
        while live:
            for i in sorted(live):
                try:
                    if streams[i] is not None:
                        with torch.cuda.stream(streams[i]):
                            p = next(gens[i])
                    else:
                        p = next(gens[i])
                    if p is not None and on_progress is not None:
                        on_progress(i, p)
                except StopIteration as e:
                    results[i] = e.value
                    live.discard(i)
    finally:
        if hinted:
            _ops.set_throughput_hint(False)
    return results


if __name__ == "__main__":
    cfgs = [TrainingConfig.from_json(a) for a in sys.argv[1:]]
    if len(cfgs) > 1:            # python -m sd_lora_trainer_amd.train a.json b.json : the jobs share this process's GPU
        res = train_concurrent(cfgs, on_progress=lambda i, p: print(f"job {i} progress {p:.2f}", flush=True))
    else:
        gen = train(cfgs[0])
        try:
            while True:
                print(f"progress {next(gen):.2f}", flush=True)
        except StopIteration as e:
            res = [e.value]
    for cfg, out in res:
        print(json.dumps({"output_save_dir": out, "job_time": cfg.job_time,
                          "images_per_second": cfg.training_attributes["images_per_second"]}))
