"""End-to-end parity on the MI355X: one full training step of the HIP path (bf16 activations, fp32
accumulation) against the fp32 oracle (oracle/unet_ref.py + loss_ref.py autograd on CPU), same seeded
inputs, tiny topologies with the exact SD1.5 / SDXL wiring.  Also: hipGraph replay == eager.

Stated tolerances (bf16 storage of every activation, ~40 layers deep):
  prediction  max-abs error <= 4e-2 * max|pred|        loss      rel error <= 2e-2
  LoRA grads / d ctx:  cosine similarity >= 0.99 and relative L2 error <= 8e-2 (whole-arena)
"""
import pytest
import torch

from oracle import loss_ref as L
from oracle import unet_ref as U

pytestmark = pytest.mark.gpu


def _inputs(cfg, B, h, seed=3):
    g = torch.Generator().manual_seed(seed)
    latent = torch.randn(B, 4, h, h, generator=g) * cfg["scaling_factor"]
    noise = torch.randn(B, 4, h, h, generator=g)
    mask = (torch.rand(B, 1, h, h, generator=g) * 0.95 + 0.05).repeat(1, 4, 1, 1).contiguous()
    t = torch.tensor([10, 900, 500, 999][:B])
    ctx = torch.randn(B, 77, cfg["cross_dim"], generator=g)
    pooled = tid = add = None
    if cfg["addition"]:
        pooled = torch.randn(B, cfg["proj_class_in"] - 6 * cfg["addition_time_embed_dim"], generator=g)
        tid = torch.tensor([[1024., 1024, 0, 0, 8. * h, 8. * h]] * B)
        add = {"text_embeds": pooled, "time_ids": tid}
    return latent, noise, mask, t, ctx, pooled, tid, add


def _oracle(cfg, sd, lora, latent, noise, t, mask, ctx, add, gamma):
    params, lg = [], {}
    for k, (A, Bm) in lora.items():
        A, Bm = A.clone().requires_grad_(True), Bm.clone().requires_grad_(True)
        lg[k] = (A, Bm)
        params += [A, Bm]
    ctx = ctx.clone().requires_grad_(True)
    acp = L.ddpm_alphas_cumprod()
    noisy = L.add_noise(acp, latent, noise, t)
    pred = U.unet_forward(cfg, sd, noisy, t, ctx, add, lora=lg)
    loss = L.diffusion_loss(pred, noise, noisy, mask, acp, t, snr_gamma=gamma)
    grads = torch.autograd.grad(loss, params + [ctx])
    return pred.detach(), float(loss), {k: (grads[2 * i], grads[2 * i + 1]) for i, k in enumerate(lora)}, grads[-1]


def _flat(d):
    return torch.cat([t.reshape(-1).float() for k in d for t in d[k]])


def _cos_rel(a, b):
    a, b = a.reshape(-1).double(), b.reshape(-1).double()
    return float(a @ b / (a.norm() * b.norm() + 1e-30)), float((a - b).norm() / (b.norm() + 1e-30))


@pytest.mark.parametrize("version,B,gamma,rank", [("tiny15", 2, 5.0, 4), ("tinyxl", 1, 5.0, 16), ("tinyxl", 2, 0.0, 8), ("tiny15", 1, 5.0, 32)])   # rank 32: padded rank 32, member-wise dX of the stacked projections
def test_step_matches_fp32_oracle(version, B, gamma, rank):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import sd_lora_trainer_amd.step as S
    import sd_lora_trainer_amd.unet as M
    from sd_lora_trainer_amd import topology
    cfg, h = U.CONFIGS[version], 16
    sd = U.init_unet_state(cfg, seed=0)
    lora = U.init_lora(cfg, rank, seed=1, b_std=0.05)
    latent, noise, mask, t, ctx, pooled, tid, add = _inputs(cfg, B, h)
    pred_o, loss_o, grads_o, gctx_o = _oracle(cfg, sd, lora, latent, noise, t, mask, ctx, add, gamma)

    rt = M.Runtime("cuda:0", B)
    unet = M.UNet(rt, topology.CONFIGS[version], sd, lora_rank=rank)
    unet.arena.load(lora)
    ts = S.TrainStep(rt, unet, latent_hw=(h, h), snr_gamma=gamma, l1_penalty=0.03, weight_decay=0.004)
    dv = lambda x: x.cuda() if x is not None else None  # noqa: E731
    ts.set_batch(dv(latent), dv(noise), dv(t), dv(mask), dv(ctx), dv(pooled), dv(tid))
    pred = ts.forward_backward().float().cpu().reshape(B, h, h, 4).permute(0, 3, 1, 2)
    torch.cuda.synchronize()
    assert torch.isfinite(pred).all()
    err = float((pred - pred_o).abs().max()) / float(pred_o.abs().max())
    assert err <= 4e-2, f"prediction error {err}"
    assert abs(float(ts.loss) - loss_o) <= 2e-2 * abs(loss_o), (float(ts.loss), loss_o)
    cos, rel = _cos_rel(_flat(unet.arena.export("grads")), _flat(grads_o))
    assert cos >= 0.99 and rel <= 8e-2, f"LoRA grads cos {cos} rel {rel}"
    gctx = ts.dctx.float().cpu().view(B, M.CTX_PAD, -1)
    cos, rel = _cos_rel(gctx[:, :77], gctx_o)
    assert cos >= 0.99 and rel <= 8e-2, f"dctx cos {cos} rel {rel}"
    assert float(gctx[:, 77:].abs().max()) == 0.0

    # hipGraph capture: replaying the captured step from the same state reproduces the eager gradients - every reduction on the
    # path (GroupNorm statistics, split-K, cross-attention dK/dV slabs, column sums) runs in a fixed order, so "reproduces" is
    # bitwise up to nothing: the tolerance below is 5e4 x tighter than round 1's (which had float atomics in the GroupNorms).
    g_eager = unet.arena.grads.clone()
    ts.capture(warmup=1)
    unet.arena.grads.zero_()
    ts.run(1e-3)
    torch.cuda.synchronize()
    cos, rel = _cos_rel(unet.arena.grads, g_eager)
    assert cos >= 0.999999 and rel <= 1e-6, f"graph replay vs eager gradients: cos {cos} rel {rel}"
    losses = []
    for i in range(5):
        ts.run(1e-3)
        losses.append(ts.total_loss())
    assert all(torch.isfinite(torch.tensor(losses)))
    assert losses[-1] < losses[0], f"loss did not go down over 5 replayed steps on a fixed batch: {losses}"


def test_prodigy_step_in_graph():
    """unet_optimizer_type = "prodigy": the captured step (d, k, numerator live on the device) trains a fixed batch, and d
    leaves d0; the group read-out of `get_current_lr` (optimizer.py:206-234) follows."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import sd_lora_trainer_amd.step as S
    import sd_lora_trainer_amd.unet as M
    from sd_lora_trainer_amd import topology
    version, B, rank, h = "tinyxl", 2, 8, 16
    cfg = U.CONFIGS[version]
    sd = U.init_unet_state(cfg, seed=0)
    lora = U.init_lora(cfg, rank, seed=1, b_std=0.05)
    latent, noise, mask, t, ctx, pooled, tid, add = _inputs(cfg, B, h)
    rt = M.Runtime("cuda:0", B)
    unet = M.UNet(rt, topology.CONFIGS[version], sd, lora_rank=rank)
    unet.arena.load(lora)
    ts = S.TrainStep(rt, unet, latent_hw=(h, h), optimizer="prodigy", prodigy_growth_rate=1.5)
    dv = lambda x: x.cuda() if x is not None else None  # noqa: E731
    ts.set_batch(dv(latent), dv(noise), dv(t), dv(mask), dv(ctx), dv(pooled), dv(tid))
    p_start = unet.arena.params.clone()
    ts.capture(warmup=1)
    assert torch.equal(unet.arena.params, p_start) and ts.prodigy.group(1.0)["k"] == 0        # capture is not training
    losses = []
    for i in range(40):
        ts.run(1.0)
        losses.append(float(ts.loss))
    grp = ts.prodigy.group(1.0)
    assert grp["k"] == 40 and grp["d"] > 10 * grp["d0"], grp
    assert torch.equal(ts.prodigy.p0, p_start)
    assert all(x == x for x in losses) and max(losses[-5:]) < 1.05 * losses[0], losses      # adapts d without blowing up


@pytest.mark.parametrize("extra", [{}, dict(is_lora=False, disable_ti=True, unet_optimizer_type="AdamW8bit", unet_lr=2e-4), dict(text_encoder_lora_optimizer="adamw", text_encoder_lora_lr=1e-4, text_encoder_lora_rank=8, ti_optimizer="prodigy", token_warmup_steps=5, training_attributes={"gpt_description": "a synthetic concept"},
                                               cond_reg_w=1e-4, tok_cov_reg_w=1.0, tok_cond_reg_w=1e-4, gradient_accumulation_steps=2),
                                   dict(tok_cond_reg_w=1e-4, cond_reg_w=1e-4), dict(use_dora=True, lora_rank=16),
                                   dict(disable_ti=True, text_encoder_lora_optimizer="adamw", text_encoder_lora_lr=1e-4, text_encoder_lora_rank=8)])
def test_train_generator_on_gpu(tmp_path, monkeypatch, extra):
    """main.py-style driver: config -> train() generator -> kohya checkpoint, on the HIP path with hipGraph replay.
    Second case: text-encoder LoRA (a21) next to the UNet LoRA, Prodigy on the token rows (a17)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import json
    import os
    monkeypatch.chdir(tmp_path)
    from sd_lora_trainer_amd.config import TrainingConfig
    from sd_lora_trainer_amd.train import train
    kw = dict(lora_training_urls="synthetic:8", concept_mode="object", pretrained_model={"path": "synthetic:tinyxl"}, seed=3, resolution=256,
              train_batch_size=2, max_train_steps=40, lora_rank=8, unet_lr=2e-3, ti_lr=2e-3, caption_dropout=0.1)
    kw.update(extra)
    cfg = TrainingConfig(**kw)
    gen = train(cfg)
    try:
        while True:
            next(gen)
    except StopIteration as e:
        config, out = e.value
    ta = json.load(open(os.path.join(out, "training_args.json")))
    tot = ta["training_attributes"]["losses"]["tot_loss"]
    assert all(map(lambda x: x == x and abs(x) < 1e4, tot)) and len(tot) >= 10
    assert sum(tot[-5:]) / 5 < sum(tot[:5]) / 5, tot          # trains on the synthetic concept
    if extra.get("is_lora") is False:      # full fine-tune (full_finetuning_example.json): the whole UNet under its diffusers names
        from safetensors.torch import load_file
        sd = load_file(os.path.join(out, "diffusion_pytorch_model.safetensors"))
        assert sd["conv_in.weight"].shape == (64, 4, 3, 3) or sd["conv_in.weight"].dim() == 4
        assert "mid_block.attentions.0.transformer_blocks.0.attn1.to_q.weight" in sd and "conv_norm_out.bias" in sd
        assert sd["conv_in.weight"].dtype == torch.bfloat16                       # saved in weight_type, like unet.save_pretrained of a bf16 UNet
        cj = json.load(open(os.path.join(out, "config.json")))                    # ... with its config.json (checkpoint.py:210-212)
        assert cj["_class_name"] == "UNet2DConditionModel" and cj["addition_embed_type"] == "text_time" and cj["cross_attention_dim"] == 128
        return
    lora_files = [n for n in os.listdir(out) if n.endswith("_sdxl_lora.safetensors") or n.endswith("_tinyxl_lora.safetensors")]
    assert lora_files
    if extra.get("token_warmup_steps"):
        w0, w1 = ta["training_attributes"]["token_warmup_losses"]        # a20 ran and its objective went down
        assert w1 < w0, (w0, w1)
        from safetensors.torch import load_file
        keys = list(load_file(os.path.join(out, lora_files[0])))
        assert any(k.startswith("lora_te1_") for k in keys) and any(k.startswith("lora_te2_") for k in keys) and any(k.startswith("lora_unet_") for k in keys)


def test_gradient_accumulation_graphs():
    """gradient_accumulation_steps = 2 on the HIP path with hipGraph replays: a micro graph (gradients only) and the boundary
    graph (accumulate, hand back, optimizer); parameters move on every second call only."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import sd_lora_trainer_amd.step as S
    import sd_lora_trainer_amd.unet as M
    from sd_lora_trainer_amd import topology
    version, B, rank, h = "tinyxl", 1, 8, 16
    cfg = U.CONFIGS[version]
    sd = U.init_unet_state(cfg, seed=0)
    latent, noise, mask, t, ctx, pooled, tid, add = _inputs(cfg, B, h)
    rt = M.Runtime("cuda:0", B)
    unet = M.UNet(rt, topology.CONFIGS[version], sd, lora_rank=rank)
    unet.arena.load(U.init_lora(cfg, rank, seed=1, b_std=0.05))
    ts = S.TrainStep(rt, unet, latent_hw=(h, h), grad_accum=2)
    dv = lambda x: x.cuda() if x is not None else None  # noqa: E731
    ts.set_batch(dv(latent), dv(noise), dv(t), dv(mask), dv(ctx), dv(pooled), dv(tid))
    ts.capture(warmup=1)
    moved = []
    for i in range(4):
        p = unet.arena.params.clone()
        ts.run(1e-3)
        torch.cuda.synchronize()
        moved.append(not torch.equal(p, unet.arena.params))
    assert moved == [False, True, False, True] and ts.opt_step == 2
    # same batch twice: the accumulated gradient equals the single-step gradient (two halves)
    g_acc = unet.arena.grads.clone()
    ts2 = S.TrainStep(rt, unet, latent_hw=(h, h), grad_accum=1)
    ts2.set_batch(dv(latent), dv(noise), dv(t), dv(mask), dv(ctx), dv(pooled), dv(tid))
    unet.arena.params.copy_(p)
    unet.arena.refresh_shadows()
    ts2.forward_backward()
    torch.cuda.synchronize()
    cos, rel = _cos_rel(g_acc, unet.arena.grads)
    assert cos >= 0.995 and rel <= 5e-2, (cos, rel)


def test_train_concurrent_on_gpu(tmp_path, monkeypatch):
    """Two independent jobs in one process, each on its own stream with its own hipGraph (train.train_concurrent): both train
    and write their own checkpoints; the graphs replay concurrently, so they must not share any scratch (split-K workspace)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import json
    import os
    monkeypatch.chdir(tmp_path)
    from sd_lora_trainer_amd.config import TrainingConfig
    from sd_lora_trainer_amd.train import train_concurrent
    cfgs = [TrainingConfig(lora_training_urls="synthetic:8", concept_mode="object", pretrained_model={"path": "synthetic:tinyxl"}, seed=3 + i, name=f"job{i}",
                           output_dir=str(tmp_path / f"out{i}"), resolution=256, train_batch_size=1, max_train_steps=40, lora_rank=8 * (i + 1),
                           unet_lr=2e-3, ti_lr=2e-3) for i in range(2)]
    res = train_concurrent(cfgs)
    for i, (cfg, out) in enumerate(res):
        tot = json.load(open(os.path.join(out, "training_args.json")))["training_attributes"]["losses"]["tot_loss"]
        assert all(x == x and abs(x) < 1e4 for x in tot) and sum(tot[-5:]) / 5 < sum(tot[:5]) / 5, (i, tot)
        assert str(tmp_path / f"out{i}") in out


def test_bench_json_contract():
    """bench.py prints ONE JSON line with the driver's fields plus `roofline`, `cpu_baseline` and the extra `two_jobs_per_gpu`
    object (toy topology so that the whole script takes seconds)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--config", "tinyxl", "--res", "256", "--steps", "3", "--warmup", "1"],
                         capture_output=True, text=True, timeout=600, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["unit"] == "images/s" and d["data"] == "synthetic" and d["dtype"] == "bf16" and "workload" in d["config"] and d["config"]["jobs_per_gpu"] == 1
    assert abs(d["value"] - d["config"]["global_batch"] * 3 / (d["ms_per_step"] * 3e-3)) <= 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and "traffic" in r
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c and len(c["step_seconds"]) >= 1
    assert d["train_loop"]["value"] > 0 and d["train_loop"]["unit"] == "images/s"          # the train() generator's own loop, extra object
    assert d["two_jobs_per_gpu"]["value"] > 0 and d["two_jobs_per_gpu"]["unit"] == "images/s"


def test_cfg1_style_sd15_job_on_the_real_topology(tmp_path, monkeypatch):
    """BASELINE configs[0]: train_configs/training_args_style_sd15.json with SURVEY 8d's overrides (512 px, batch 1, rank 4, 50 steps) - the
    reference's CPU plumbing config - as a whole job on the REAL SD1.5 topology through train(): style mode, textual inversion on,
    checkpoints at the reference's cadence, kohya file with all 150 adapters.  (Random-init weights and a synthetic latent cache: there
    are no checkpoints or datasets offline; max_train_steps < 100 also exercises the guard of main.py:457's modulo.)"""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import json
    import os
    monkeypatch.chdir(tmp_path)
    from safetensors.torch import load_file
    from sd_lora_trainer_amd.config import TrainingConfig
    from sd_lora_trainer_amd.train import train
    cfg = TrainingConfig(name="twisting_realities_sd15", sd_model_version="sd15", lora_training_urls="synthetic:6", concept_mode="style",
                         sample_imgs_lora_scale=0.8, seed=0, resolution=512, train_batch_size=1, n_sample_imgs=8, max_train_steps=50,
                         checkpointing_steps=200, disable_ti=False, caption_model="blip", ti_lr=0.001, unet_lr=0.0003, lora_rank=4, debug=True,
                         pretrained_model={"path": "synthetic:sd15"})
    gen = train(cfg)
    n = 0
    try:
        while True:
            next(gen)
            n += 1
    except StopIteration as e:
        config, out = e.value
    assert n == 51                                          # max_train_steps + 1 steps (main.py:462)
    ta = json.load(open(os.path.join(out, "training_args.json")))
    tot = ta["training_attributes"]["losses"]["tot_loss"]
    assert all(x == x and abs(x) < 1e4 for x in tot) and len(tot) >= 10
    assert ta["concept_mode"] == "style" and ta["lora_rank"] == 4 and ta["pretrained_model"]["version"] == "sd15"
    sd = load_file(os.path.join(out, "twisting_realities_sd15_sd15_lora.safetensors"))
    downs = [k for k in sd if k.endswith(".lora_down.weight")]
    assert len(downs) == 150 and all(sd[k].shape[0] == 4 for k in downs)            # 128 linear + 22 conv adapters
    emb = load_file(os.path.join(out, "twisting_realities_sd15_sd15_embeddings.safetensors"))
    assert set(emb) == {"clip_l"} and tuple(emb["clip_l"].shape) == (3, 768)


@pytest.mark.parametrize("name,kw", [
    ("cfg2", dict(sd_model_version="sd15", concept_mode="face", resolution=512, train_batch_size=4, lora_rank=16, pretrained_model={"path": "synthetic:sd15"})),
    ("cfg3", dict(sd_model_version="sdxl", concept_mode="object", resolution=1024, train_batch_size=1, lora_rank=16, pretrained_model={"path": "synthetic:sdxl"})),
    ("sweep", dict(sd_model_version="sd15", concept_mode="style", resolution=512, train_batch_size=8, lora_rank=24, use_dora=True, disable_ti=True,
                   unet_lr=0.001, pretrained_model={"path": "synthetic:sd15"}))])
def test_baseline_config_jobs_on_the_real_topologies(tmp_path, monkeypatch, name, kw):
    """BASELINE configs[1] (SD1.5 512 px face, rank 16 + TI, batch 4), configs[2] (SDXL 1024 px object, rank 16 + TI, batch 1 - the headline) and one
    draw of configs[3]'s sweep (create_hyperparam_sweep.py:52-90: SD1.5, batch 8, rank 24, DoRA, no TI) as whole jobs through train() on the
    real topologies: 40 optimizer steps under hipGraph replay, the loss of the synthetic concept falls, checkpoint files complete."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import json
    import os
    monkeypatch.chdir(tmp_path)
    from safetensors.torch import load_file
    from sd_lora_trainer_amd.config import TrainingConfig
    from sd_lora_trainer_amd.train import train
    base = dict(name=name, lora_training_urls="synthetic:8", seed=0, max_train_steps=40, n_sample_imgs=0, unet_lr=1e-3, ti_lr=1e-3)
    base.update(kw)
    cfg = TrainingConfig(**base)
    gen = train(cfg)
    try:
        while True:
            next(gen)
    except StopIteration as e:
        config, out = e.value
    ta = json.load(open(os.path.join(out, "training_args.json")))
    tot = ta["training_attributes"]["losses"]["tot_loss"]
    assert all(x == x and abs(x) < 1e4 for x in tot) and len(tot) >= 10
    assert sum(tot[-5:]) / 5 < sum(tot[:5]) / 5, tot
    ver = kw["sd_model_version"]
    sd = load_file(os.path.join(out, f"{name}_{ver}_lora.safetensors"))
    n_ad = 577 if ver == "sdxl" else 150
    assert len([k for k in sd if k.endswith(".lora_down.weight")]) == n_ad
    assert len([k for k in sd if k.endswith(".dora_scale")]) == (n_ad if kw.get("use_dora") else 0)
    assert ta["training_attributes"].get("images_per_second", 1.0) > 0
