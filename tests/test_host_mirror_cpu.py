"""Host-side mirrors of the reference's interfaces (CPU): TrainingConfig vs snapshots of the reference's own class on
its shipped train_configs/*.json, LR schedules, caption dropout, checkpoint format, and the train() generator driven
end-to-end through the op emulation on a tiny synthetic job."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import loss_ref as L
from tests import emu_ops

import sd_lora_trainer_amd.unet as unet_mod
from sd_lora_trainer_amd import checkpoint as ckpt
from sd_lora_trainer_amd import schedule, topology
from sd_lora_trainer_amd.config import TrainingConfig


def test_training_config_matches_reference_snapshots(golden_dir, tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    snaps = json.load(open(os.path.join(golden_dir, "config_snapshots.json")))
    assert len(snaps) == 11
    for fname, s in snaps.items():
        c = TrainingConfig(**s["input"])
        d = c.model_dump()
        for k, ref in s["derived"].items():
            if k == "pretrained_model":
                assert (d[k] or {}).get("version") == (ref or {}).get("version"), (fname, k)
                continue
            assert d[k] == ref, (fname, k, d[k], ref)
        assert os.path.isdir(c.output_dir) and c.device == "cuda:0"
        p = tmp_path / "roundtrip.json"
        c.save_as_json(str(p))
        assert json.load(open(p))["lora_rank"] == c.lora_rank


def test_lr_schedules_and_caption_dropout(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    c = TrainingConfig(lora_training_urls="x", concept_mode="object", sd_model_version="sdxl", max_train_steps=400, unet_lr=1e-3, ti_lr=1e-3)
    assert c.unet_lr_warmup_steps == 400
    spe, epochs = 10, 40
    for gs in (0, 1, 137, 279, 281, 399):
        ep, st = divmod(gs, spe)
        f = schedule.completion_fraction(ep, st, spe, epochs)
        got = schedule.learning_rates(c, gs, f)
        lr_u, lr_t, f_ref = L.lr_schedule(gs, st, ep, spe, epochs, unet_lr=1e-3, unet_lr_warmup_steps=400, ti_lr=1e-3)
        assert abs(f - f_ref) < 1e-12 and abs(got["unet"] - lr_u) < 1e-15 and abs(got["textual_inversion"] - lr_t) < 1e-15
    assert schedule.learning_rates(c, 0, 0.0)["unet"] == pytest.approx(5e-5)           # LoRA + TI cold start (main.py:236)
    assert schedule.learning_rates(c, 400, 1.0)["unet"] == pytest.approx(1e-3)
    assert schedule.learning_rates(c, 300, 0.75)["textual_inversion"] == 0.0            # frozen after 0.7
    c2 = TrainingConfig(lora_training_urls="x", concept_mode="style", sd_model_version="sd15", disable_ti=True)
    assert schedule.base_unet_lr(c2.is_lora, c2.disable_ti) == 2e-4 and schedule.base_unet_lr(False, True) == 1e-5
    np.random.seed(0)
    caps = schedule.apply_caption_dropout(["a"] * 1000, 0.1, "<s0><s1><s2>")
    assert 60 < sum(x == "<s0><s1><s2>" for x in caps) < 140
    assert schedule.apply_caption_dropout(["a", "b"], 0.0, "T") == ["a", "b"]


def test_checkpoint_format_roundtrip(tmp_path):
    cfg = topology.CONFIGS["tinyxl"]
    rt = unet_mod.Runtime("cpu", 1, act_dtype=torch.float32, ops=emu_ops)
    from oracle import unet_ref as U
    unet = unet_mod.UNet(rt, cfg, U.init_unet_state(U.CONFIGS["tinyxl"], seed=0), lora_rank=4)
    lora = U.init_lora(U.CONFIGS["tinyxl"], 4, seed=1, b_std=0.05)
    unet.arena.load(lora)
    rows = [torch.randn(3, 64), torch.randn(3, 64)]
    files = ckpt.save_checkpoint(str(tmp_path), 10, unet.arena, rows, {"TOK": "<s0><s1><s2>"}, "my concept.v1", "sdxl")
    assert os.path.basename(files["lora"]) == "my_concept_v1_sdxl_lora.safetensors"
    assert os.path.basename(files["embeddings"]) == "my_concept_v1_sdxl_embeddings.safetensors"
    from safetensors.torch import load_file
    sd = load_file(files["lora"])
    targets = topology.lora_targets(cfg)
    assert len(sd) == 3 * len(targets)
    k = "lora_unet_down_blocks_1_attentions_0_transformer_blocks_0_attn1_to_q"
    assert sd[k + ".lora_down.weight"].shape == (4, 128) and sd[k + ".lora_up.weight"].shape == (128, 4) and float(sd[k + ".alpha"]) == 4.0
    # byte fidelity (checkpoint.py:84-102, 183-209 of the reference): alpha = torch.tensor(len(lora_down)) -> 0-dim int64; the adapter
    # tensors and the token rows leave in the training dtype (weight_type, bf16 by default)
    assert sd[k + ".alpha"].dtype == torch.int64 and sd[k + ".alpha"].dim() == 0
    assert sd[k + ".lora_down.weight"].dtype == torch.bfloat16 and sd[k + ".lora_up.weight"].dtype == torch.bfloat16
    kc = "lora_unet_down_blocks_0_resnets_0_conv2"
    assert sd[kc + ".lora_down.weight"].shape == (4, 64, 3, 3) and sd[kc + ".lora_up.weight"].shape == (64, 4, 1, 1)
    assert not any("base_model" in key for key in sd)
    back = ckpt.load_lora(files["lora"], targets)
    for m in targets:
        torch.testing.assert_close(back[m][0], lora[m][0], rtol=8e-3, atol=1e-4)      # bf16 on disk
        torch.testing.assert_close(back[m][1], lora[m][1], rtol=8e-3, atol=1e-4)
    emb = ckpt.load_embeddings(files["embeddings"])
    assert emb[0].dtype == torch.bfloat16 and torch.equal(emb[0], rows[0].to(torch.bfloat16)) and torch.equal(emb[1], rows[1].to(torch.bfloat16))
    assert json.load(open(tmp_path / "special_params.json")) == {"TOK": "<s0><s1><s2>"}
    assert json.load(open(tmp_path / "adapter_config.json"))["r"] == 4


@pytest.mark.parametrize("version,disable_ti", [("tinyxl", False), ("tiny15", True)])
def test_train_generator_end_to_end(tmp_path, monkeypatch, version, disable_ti):
    monkeypatch.chdir(tmp_path)
    from sd_lora_trainer_amd.train import train
    cfg = TrainingConfig(lora_training_urls="synthetic:4", concept_mode="object", pretrained_model={"path": f"synthetic:{version}"}, seed=1,
                         resolution=256 if version == "tinyxl" else 128, train_batch_size=2, max_train_steps=3, lora_rank=4, disable_ti=disable_ti,
                         unet_lr=1e-3, ti_lr=1e-3, caption_dropout=0.5, tok_cond_reg_w=0.0 if disable_ti else 1e-3)
    rt = unet_mod.Runtime("cpu", 2, act_dtype=torch.float32, ops=emu_ops)
    gen = train(cfg, runtime=rt)
    progress = []
    try:
        while True:
            progress.append(next(gen))
    except StopIteration as e:
        config, out_dir = e.value
    assert len(progress) == 4 and progress[-1] == 1.0          # max_train_steps + 1 steps (main.py:462), no ZeroDivisionError (:457)
    # main.py:467-470: fewer than 27 steps since the last save (none: last_save_step = 0) -> the final directory is checkpoint-{last_save_step}
    assert os.path.isdir(out_dir) and out_dir.endswith("checkpoint-0")
    names = sorted(os.listdir(out_dir))
    assert any(n.endswith("_lora.safetensors") for n in names) and "training_args.json" in names and "special_params.json" in names
    assert any(n.endswith("_embeddings.safetensors") for n in names)          # checkpoint.py:153-158: written even when the rows were not trained
    ta = json.load(open(os.path.join(out_dir, "training_args.json")))
    assert ta["num_train_epochs"] == 2 and ta["pretrained_model"]["version"] == version
    assert np.isfinite(ta["training_attributes"]["losses"]["tot_loss"]).all()


def test_latent_cache_matches_reference_dataset(golden_dir):
    """dataset.py:31-193 (PreprocessedDataset) run in this container on a 3-image dataset with a duck-typed VAE encoder
    (tests/golden/dataset.pt, oracle/gen_golden.py::gen_dataset): processed captions, latent-resolution masks and the
    per-fetch posterior samples of `sd_lora_trainer_amd.dataset.LatentCache`."""
    import os
    import torch
    from PIL import Image
    from sd_lora_trainer_amd.dataset import LatentCache
    g = torch.load(os.path.join(golden_dir, "dataset.pt"), weights_only=False)
    caps = list(g["captions_in"])
    caps[g["nan_index"]] = float("nan")
    cache = LatentCache(g["posteriors"], [Image.fromarray(m) for m in g["masks_u8"]], caps, scaling_factor=g["scaling_factor"],
                        size=g["size"], substitute_caption_map=g["substitute"])
    assert cache.captions == list(g["captions"])
    assert "<s0><s1><s2>" in cache.captions[0] and cache.captions[1] == "" and cache.captions[2].count("<s0><s1><s2>") == 2
    torch.manual_seed(g["fetch_seed"])          # the reference samples from the global RNG
    for i in range(len(cache)):
        c, lat, m = cache[i]
        torch.testing.assert_close(lat, g["latents"][i], rtol=1e-6, atol=1e-7)
        assert torch.equal(m, g["masks"][i]) and m.shape == lat.shape
    a, b = cache[0][1], cache[0][1]
    assert not torch.equal(a, b)                # a fresh posterior sample on every fetch (dataset.py:186)
    caps_b, lats, masks = cache.batch([2, 0])
    assert lats.shape == (2, 4, 6, 8) and masks.shape == lats.shape and caps_b[0] == cache.captions[2]
    ones = LatentCache(g["posteriors"][:1], None, ["x"], scaling_factor=1.0, size=g["size"])
    assert torch.equal(ones.masks[0], torch.ones(4, 6, 8))


def test_train_concurrent_and_sweep_groups(tmp_path, monkeypatch):
    """Two jobs advanced in lock-step in one process (train.train_concurrent; on the GPU each on its own stream / graph), and
    the sweep launcher's grouping of configs per GPU process."""
    monkeypatch.chdir(tmp_path)
    from sd_lora_trainer_amd import parallel
    from sd_lora_trainer_amd.train import train_concurrent
    cfgs = [TrainingConfig(lora_training_urls="synthetic:4", concept_mode="object", pretrained_model={"path": "synthetic:tiny15"}, seed=1 + i, name=f"job{i}",
                           output_dir=str(tmp_path / f"out{i}"), resolution=128, train_batch_size=1, max_train_steps=3 + i, lora_rank=4, disable_ti=True,
                           unet_lr=1e-3) for i in range(2)]
    rts = [unet_mod.Runtime("cpu", 1, act_dtype=torch.float32, ops=emu_ops) for _ in cfgs]
    seen = []
    res = train_concurrent(cfgs, on_progress=lambda i, p: seen.append(i), runtimes=rts)
    assert [os.path.basename(out) for _, out in res] == ["checkpoint-0", "checkpoint-0"]      # the reference's final-directory rule (main.py:467-470)
    assert all(any(n.endswith("_lora.safetensors") for n in os.listdir(out)) for _, out in res)
    assert str(tmp_path / "out0") in res[0][1] and str(tmp_path / "out1") in res[1][1]
    assert set(seen) == {0, 1} and seen[:2] == [0, 1]                       # interleaved, not one after the other
    plan = parallel.sweep_plan([f"c{i}" for i in range(10)], 2, jobs_per_gpu=2)
    assert [(g, w) for _, g, w in plan] == [(0, 0), (0, 0), (1, 0), (1, 0), (0, 1), (0, 1), (1, 1), (1, 1), (0, 2), (0, 2)]
    launched = parallel.run_sweep([f"c{i}.json" for i in range(5)], 2, dry_run=True, jobs_per_gpu=2)
    assert [cmd[-2:] if len(cmd) > 4 else cmd[-1:] for cmd, _ in launched] == [["c0.json", "c1.json"], ["c2.json", "c3.json"], ["c4.json"]]
    assert [env["HIP_VISIBLE_DEVICES"] for _, env in launched] == ["0", "1", "0"]
