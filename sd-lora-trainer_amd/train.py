"""`train(config)`: the reference's orchestrator interface (main.py:34-551) for the part this engine owns - a *generator*
that yields progress floats and returns `(config, output_save_dir)` through StopIteration.value, driven by the same
train_configs/*.json files:   python -m sd_lora_trainer_amd.train cfg.json

What is NOT here (out of scope): preprocessing/captioning/masking, VAE latent caching, validation rendering.
The data source is therefore a *latent cache* - either a `.pt` file with the tensors the reference's dataset would produce
(`latents [N,4,h,w]`, `masks [N,4,h,w]`, `input_ids [N,77]` per tokenizer, `token_lists`) or, when `lora_training_urls`
starts with "synthetic:", a seeded synthetic cache.  Model weights come from diffusers/HF-named state dicts (`.safetensors` /
`.pt`) or, for `pretrained_model = {"path": "synthetic:<version>"}`, seeded random weights of the exact architecture.
"""
import json
import math
import os
import sys
import time

import numpy as np
import torch

from . import checkpoint as ckpt
from . import schedule, topology
from .config import TrainingConfig
from .embedding_handler import TokenEmbeddingsHandler
from .dataset import DiagonalGaussian
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


def build_models(config, rt):
    """-> (unet, text_stack or None, version).  Mirrors load_models (trainer/models.py:7-54) for the parts on the step."""
    from . import clip as CL
    from . import step as S
    from . import unet as M
    path = (config.pretrained_model or {}).get("path", "")
    if path.startswith("synthetic:"):
        version = path.split(":", 1)[1]
        cfg = topology.CONFIGS[version]
        sd = _random_state(topology.param_shapes(cfg), rt.device, seed=config.seed)
    else:
        from safetensors.torch import load_file
        sd = load_file(path)
        version = config.sd_model_version or ("sdxl" if "add_embedding.linear_1.weight" in sd else "sd15")
        cfg = topology.CONFIGS[version]
    if config.is_lora:
        unet = M.UNet(rt, cfg, sd, lora_rank=config.lora_rank, lora_alpha_multiplier=config.lora_alpha_multiplier)
    else:                          # main.py:144-149: full fine-tune, every UNet parameter trained
        from . import fullft
        unet = M.UNet(rt, cfg, sd, trainer=fullft.WeightTrainer(rt))
    text = None
    if config.text_encoder_lora_optimizer is not None and config.disable_ti:
        raise NotImplementedError("text-encoder LoRA without textual inversion: the text stack is only built for TI runs")
    if not config.disable_ti:
        tiny = version.startswith("tiny")
        kinds = (["tiny_l", "tiny_g"] if tiny else ["clip_l", "clip_g"]) if cfg["addition"] else (["tiny_l"] if tiny else ["clip_l"])
        encs = []
        te_arena = None
        if config.text_encoder_lora_optimizer is not None:         # a21 (main.py:116-126, optimizer.py:157-202)
            te_arena = M.LoraArena(rt, config.text_encoder_lora_rank, config.lora_alpha_multiplier, problems=[])
        for i, kd in enumerate(kinds):
            c = topology.CLIP_CONFIGS[kd]
            csd = _random_state(topology.clip_param_shapes(c, config.n_tokens), rt.device, seed=config.seed + 1 + i)
            encs.append(CL.ClipTextEncoder(rt, f"te{i + 1}", csd, heads=c["heads"], act=c["act"], mode="penultimate" if cfg["addition"] else "last",
                                           with_projection=bool(c["proj"]), n_train=config.n_tokens, arena=te_arena,
                                           lora_prefix="text_encoder." if i == 0 else "text_encoder_2."))
        if te_arena is not None:
            te_arena.finalize()
        text = S.TextStack(rt, encs, pool_mode="argmax", arena=te_arena)
    return unet, text, version


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


def train(config: TrainingConfig, runtime=None, every_step=False):
    """Generator: yields progress in [0,1]; returns (config, output_save_dir).  every_step: also yield (None) after every
    optimizer call, so that several jobs can be advanced in lock-step by train_concurrent."""
    from . import step as S
    from . import unet as M
    np.random.seed(config.seed)
    torch.manual_seed(config.seed)
    B = config.train_batch_size
    rt = runtime or M.Runtime(config.device, B)
    unet, text, version = build_models(config, rt)
    cfg = unet.cfg
    config.pretrained_model = dict(config.pretrained_model or {}, version=version)
    if config.train_img_size is None:
        config.train_img_size = [config.resolution, config.resolution]
    w, h = config.train_img_size[0] // 8, config.train_img_size[1] // 8
    if config.lora_training_urls.startswith("synthetic:"):
        n_img = int(config.lora_training_urls.split(":")[1] or 8)
        vocab = text.encoders[0].V if text is not None else 49411
        cache = synthetic_cache(cfg, n_img, h, w, vocab, config.n_tokens, config.seed)
    else:
        cache = torch.load(config.lora_training_urls)
    n_img = (cache["posterior"] if "posterior" in cache else cache["latents"]).shape[0]
    steps_per_epoch = max(n_img // B, 1)
    config.num_train_epochs = math.ceil(config.max_train_steps / steps_per_epoch)      # main.py:207

    import torch.distributed as dist
    ddp = (not config.is_lora) and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    ts = S.TrainStep(rt, unet, latent_hw=(h, w), process_group=True if ddp else None, snr_gamma=config.snr_gamma, l1_penalty=config.l1_penalty, weight_decay=config.lora_weight_decay,
                     grad_accum=config.gradient_accumulation_steps, text=text, n_tokens=config.n_tokens,
                     token_attention_loss_w=config.token_attention_loss_w, ti_weight_decay=config.ti_weight_decay,
                     optimizer=config.unet_optimizer_type, ti_optimizer=config.ti_optimizer,
                     prodigy_d_coef=config.prodigy_d_coef, prodigy_growth_rate=config.unet_prodigy_growth_factor,
                     text_lora_weight_decay=config.text_encoder_lora_weight_decay,
                     cond_reg_w=config.cond_reg_w, tok_cov_reg_w=config.tok_cov_reg_w)
    handler = None
    if text is not None:
        handler = TokenEmbeddingsHandler(ts.ti, config.inserting_list_tokens)
        handler.initialize_new_tokens(seed=config.seed)
    # a20 token warm-up (main.py -> embedding_handler.pre_optimize_token_embeddings, :321-399): only with token_warmup_steps > 0
    # and a description of the concept (training_attributes["gpt_description"], tokenised by the data stage: `description_ids`)
    if text is not None and config.token_warmup_steps > 0 and (config.training_attributes or {}).get("gpt_description") \
            and cache.get("description_ids") is not None and cache.get("tok_list") is not None:
        def row(l):
            r = torch.full((77,), int(cache["input_ids"][0, -1]), dtype=torch.int64)
            r[:len(l)] = torch.tensor(l)
            return r
        n_enc = len(text.encoders)
        warm = ts.token_warmup([row(cache["tok_list"])] * n_enc, [row(cache["description_ids"])] * n_enc, config.token_warmup_steps, config.ti_lr)
        config.training_attributes = dict(config.training_attributes, token_warmup_losses=[warm[0], warm[-1]])
    g = torch.Generator(device=rt.device).manual_seed(config.seed)
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
    optimizers = OptimizerCollection(ts, config)
    checkpoint_dir = os.path.join(config.output_dir, "checkpoints")
    os.makedirs(checkpoint_dir, exist_ok=True)
    time_ids = torch.tensor([[1024., 1024, 0, 0, float(config.resolution), float(config.resolution)]] * B) if cfg["addition"] else None
    tok_string_ids = cache.get("tok_list")
    losses = {"img_loss": [], "tot_loss": []}
    global_step, images_done, start = 0, 0, time.time()
    captured = False
    perm_rng = np.random.RandomState(config.seed)
    done = False
    for epoch in range(config.num_train_epochs):
        order = perm_rng.permutation(n_img)
        for step_in_epoch in range(steps_per_epoch):
            completion_f = schedule.completion_fraction(epoch, step_in_epoch, steps_per_epoch, config.num_train_epochs)
            lrs = schedule.learning_rates(config, global_step, completion_f, ti_active=text is not None,
                                          text_lora_active=ts.te_arena is not None)
            if ts.te_arena is not None:
                optimizers.optimizers["text_encoders"].param_groups[0]["lr"] = lrs["text_encoders"]
            optimizers.optimizers["unet"].param_groups[0]["lr"] = lrs["unet"]
            if text is not None:
                optimizers.optimizers["textual_inversion"].param_groups[0]["lr"] = lrs["textual_inversion"]
            idx = torch.as_tensor(order[step_in_epoch * B:(step_in_epoch + 1) * B])
            mask = cache["masks"][idx].to(rt.device)
            if "posterior" in cache:       # dataset.py:184-187: latent_dist.sample() * scaling_factor on EVERY fetch
                dist = DiagonalGaussian(cache["posterior"][idx].to(rt.device))
                latent = dist.sample(g) * cfg["scaling_factor"]
            else:
                latent = cache["latents"][idx].to(rt.device)
            noise = torch.randn(latent.shape, generator=g, device=rt.device)
            if config.noise_offset > 0.0:                                                # main.py:313-317
                noise += config.noise_offset * torch.randn((B, 4, 1, 1), generator=g, device=rt.device)
            timesteps = torch.randint(0, 1000, (B,), generator=g, device=rt.device)
            if text is not None:
                ids = cache["input_ids"][idx].clone()
                lists = [cache["token_lists"][int(i)] for i in idx]
                if config.caption_dropout > 0.0 and tok_string_ids is not None:           # main.py:300-304
                    for b in range(B):
                        if np.random.rand() < config.caption_dropout:
                            lists[b] = list(tok_string_ids)
                            ids[b] = ids[b, -1]
                            ids[b, :len(tok_string_ids)] = torch.tensor(tok_string_ids)
                ts.set_batch(latent, noise, timesteps, mask, time_ids=time_ids, ids=[ids] * len(text.encoders), caption_token_lists=lists)
            else:
                ctx = torch.randn(B, 77, cfg["cross_dim"], generator=g, device=rt.device)
                pooled = torch.randn(B, cfg["proj_class_in"] - 6 * cfg["addition_time_embed_dim"], generator=g, device=rt.device) if cfg["addition"] else None
                ts.set_batch(latent, noise, timesteps, mask, ctx, pooled, time_ids)
            if not captured and rt.device.type == "cuda":
                ts.capture(warmup=1)
                captured = True
            optimizers.step(last_batch=step_in_epoch + 1 == steps_per_epoch)
            optimizers.zero_grad()
            if global_step % max(config.max_train_steps // 20, 1) == 0:
                losses["img_loss"].append(float(ts.loss))
                losses["tot_loss"].append(ts.total_loss())
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
    output_save_dir = os.path.join(checkpoint_dir, f"checkpoint-{global_step}")
    config.job_time = time.time() - config.start_time
    config.training_attributes = dict(config.training_attributes, images_per_second=images_done / max(time.time() - start, 1e-9), losses=losses)
    ckpt.save_checkpoint(output_save_dir, global_step, arena, ts.ti.rows if ts.ti is not None else None, config.token_dict, config.name,
                         version, config=config, text_arena=ts.te_arena, unet_weights=unet.trainer)
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
