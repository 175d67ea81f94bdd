"""train() as a job driver for a reference-style config (main.py:34-551), on CPU through the op emulation: a preprocessed image
FOLDER (captions.csv + images + masks) -> VAE-encoded latent cache, CLIP BPE tokenizer files -> ids / token lists, trigger
tokens added and initialised, checkpoint cadence (`checkpointing_steps`, also at step 0) and the final-directory rule,
`disable_ti` with the captions still encoded by the (frozen) text encoders, and the refusal of fields this engine does not build."""
import csv
import json
import os

import numpy as np
import pytest
import torch
from PIL import Image

import sd_lora_trainer_amd.unet as unet_mod
from sd_lora_trainer_amd.config import TrainingConfig
from tests import emu_ops
from tests.test_tokenizer_cpu import _train_bpe


def _dataset(tmp_path, n=5, size=64):
    d = tmp_path / "images_out"
    os.makedirs(d)
    rng = np.random.RandomState(0)
    rows = []
    for i in range(n):
        Image.fromarray(rng.randint(0, 255, (size, size, 3), dtype=np.uint8)).save(d / f"{i}.src.jpg")
        Image.fromarray(rng.randint(0, 255, (size, size), dtype=np.uint8)).save(d / f"{i}.mask.jpg")
        rows.append(dict(image_path=f"{i}.src.jpg", caption=f"a photo of TOK on the grass, number {i}" if i != 2 else "a dog in front of a house", mask_path=f"{i}.mask.jpg"))
    with open(d / "captions.csv", "w", newline="") as fh:
        w = csv.DictWriter(fh, fieldnames=["image_path", "caption", "mask_path"])
        w.writeheader()
        w.writerows(rows)
    return str(d)


def _tokenizer_dir(tmp_path):
    vocab, merges = _train_bpe(200)
    d = tmp_path / "tokenizer"
    os.makedirs(d)
    json.dump(vocab, open(d / "vocab.json", "w", encoding="utf-8"), ensure_ascii=False)
    open(d / "merges.txt", "w", encoding="utf-8").write("#version: 0.2\n" + "\n".join(f"{a} {b}" for a, b in merges) + "\n")
    return str(d), len(vocab)


def _run(gen):
    progress = []
    try:
        while True:
            progress.append(next(gen))
    except StopIteration as e:
        return progress, e.value


@pytest.mark.parametrize("version,disable_ti", [("tinyxl", False), ("tiny15", True)])
def test_train_from_folder_with_tokenizer(tmp_path, monkeypatch, version, disable_ti):
    monkeypatch.chdir(tmp_path)
    from sd_lora_trainer_amd import train as T
    data, (tok_dir, vocab_size) = _dataset(tmp_path), _tokenizer_dir(tmp_path)
    cfg = TrainingConfig(lora_training_urls=data, concept_mode="object", name="my run", seed=2, resolution=64, train_batch_size=2, max_train_steps=60,
                         checkpointing_steps=20, lora_rank=4, disable_ti=disable_ti, unet_lr=1e-3, ti_lr=1e-3, caption_dropout=0.3,
                         tok_cond_reg_w=0.0 if disable_ti else 1e-3,        # (reg captions through the tokenizer files, loss.py:241-251)
                         pretrained_model={"path": f"synthetic:{version}", "tokenizer_path": tok_dir})
    seen = {}
    real_set = None

    import sd_lora_trainer_amd.step as S
    real_set = S.TrainStep.set_batch

    def spy(self, latent, noise, timesteps, mask, ctx=None, pooled=None, time_ids=None, ids=None, caption_token_lists=None, **kw):
        # train() hands the ids over as (per-encoder tables, batch rows, pooling-position table): the batch's ids are the gathered rows
        ids_seen = [t[ids[1]] for t in ids[0]] if isinstance(ids, tuple) else ids
        seen.setdefault("calls", []).append(dict(ctx=None if ctx is None else ctx.clone(), ids=ids_seen, table=kw.get("caption_table"), mask=mask.clone(), latent=latent.clone()))
        return real_set(self, latent, noise, timesteps, mask, ctx, pooled, time_ids, ids, caption_token_lists, **kw)
    monkeypatch.setattr(S.TrainStep, "set_batch", spy)
    rt = unet_mod.Runtime("cpu", 2, act_dtype=torch.float32, ops=emu_ops)
    progress, (config, out_dir) = _run(T.train(cfg, runtime=rt))
    assert progress[-1] == 1.0 and config.num_train_epochs == 20                    # ceil(5 / 2) = 3 batches per epoch (drop_last=False)
    ck = os.path.dirname(out_dir)
    # cadence (main.py:399): steps 0 and 20 (40 is not < 60 - 25); 20 epochs x 3 batches end the loop at 60 steps; final (466-470): 60 - 20 > 26 -> checkpoint-60
    assert sorted(os.listdir(ck)) == ["checkpoint-0", "checkpoint-20", "checkpoint-60"] and out_dir.endswith("checkpoint-60")
    names = os.listdir(out_dir)
    assert f"my_run_{version}_lora.safetensors" in names and f"my_run_{version}_embeddings.safetensors" in names and "training_args.json" in names
    ta = json.load(open(os.path.join(out_dir, "training_args.json")))
    assert ta["training_attributes"]["images_per_second"] > 0 and all(np.isfinite(ta["training_attributes"]["losses"]["tot_loss"]))
    calls = seen["calls"]
    assert len(calls) == 60
    lat = torch.stack([c["latent"] for c in calls])
    assert lat.shape[1:] == (2, 4, 8, 8) and float(lat.std()) > 0
    m = calls[0]["mask"]
    assert m.shape == (2, 4, 8, 8) and torch.equal(m[:, 0], m[:, 3]) and 0.0 <= float(m.min()) and float(m.max()) <= 1.0 and float(m.std()) > 0   # latent-resolution masks from the jpgs
    tok_ids = [vocab_size, vocab_size + 1, vocab_size + 2]
    if disable_ti:
        # captions are encoded by the frozen text encoders (no random conditioning): the conditioning only takes the values of the
        # 5 captions + the caption-dropout caption
        uniq = {tuple(c["ctx"][b][3:9].flatten()[:128].tolist()) for c in calls for b in range(2)}     # (position 0 only sees BOS: causal)
        assert 2 <= len(uniq) <= 6 and all(c["ids"] is None for c in calls)
    else:
        ids = torch.cat([c["ids"][0] for c in calls])
        assert ids.shape[1] == 77 and all(len(c["ids"]) == 2 for c in calls)
        rows = [row.tolist() for c in calls for row in c["ids"][0]]
        assert any(all(t in r for t in tok_ids) for r in rows)                       # "tok" -> <s0><s1><s2> substitution reached the tokenizer
        assert any(not any(t in r for t in tok_ids) for r in rows)                   # the caption without the trigger word
        drop = [r for r in rows if r[1:4] == tok_ids and r[4] == r[-1]]              # caption dropout -> the bare trigger string (bos, 3 tokens, eos...)
        assert 0 < len(drop) < len(rows)
        # the token-attention loss's per-caption constants come from a device table, row n_img = the dropout caption
        tabs = [c["table"] for c in calls]
        assert all(t is not None and t[0][3].shape[0] == 6 for t in tabs) and any(int(x) == 5 for t in tabs for x in t[1])
        from safetensors.torch import load_file
        emb = load_file(os.path.join(out_dir, f"my_run_{version}_embeddings.safetensors"))
        e0 = load_file(os.path.join(ck, "checkpoint-0", f"my_run_{version}_embeddings.safetensors"))
        assert set(emb) == {"clip_l", "clip_g"} and not torch.equal(emb["clip_l"], e0["clip_l"])     # the token rows were trained


def test_text_encoder_lora_without_textual_inversion(tmp_path, monkeypatch):
    """`text_encoder_lora_optimizer` with `disable_ti` (main.py:116-133: the adapters are built whether or not the token rows train): the text encoders stay inside the
    step for the adapters' gradients, the token rows keep the values they were initialised with, no token-attention loss, no token regularisers in the logged total."""
    monkeypatch.chdir(tmp_path)
    from safetensors.torch import load_file
    from sd_lora_trainer_amd import train as T
    import sd_lora_trainer_amd.step as S
    cfg = TrainingConfig(lora_training_urls="synthetic:4", concept_mode="object", name="te only", seed=4, resolution=128, train_batch_size=1, max_train_steps=30,
                         checkpointing_steps=100, lora_rank=4, disable_ti=True, text_encoder_lora_optimizer="adamw", text_encoder_lora_lr=2e-2, text_encoder_lora_rank=4,
                         txt_encoders_lr_warmup_steps=0, unet_lr=1e-3, pretrained_model={"path": "synthetic:tinyxl"})
    seen = []
    real_run = S.TrainStep._run

    def spy(self, lr, lr_ti=0.0, lr_te=0.0, last_batch=False):
        seen.append((lr_ti, lr_te, self.text is not None, self.ti_trainable, self.ta_w, self.te_arena is not None))
        return real_run(self, lr, lr_ti, lr_te, last_batch)
    monkeypatch.setattr(S.TrainStep, "_run", spy)
    rt = unet_mod.Runtime("cpu", 1, act_dtype=torch.float32, ops=emu_ops)
    progress, (config, out_dir) = _run(T.train(cfg, runtime=rt))
    assert progress[-1] == 1.0 and len(seen) >= 30 and out_dir.endswith("checkpoint-31")      # (checkpoint-0, then the final one: the loop runs max_train_steps + 1 steps, main.py:462-470)
    assert all(s[0] == 0.0 and s[2] and not s[3] and s[4] == 0.0 and s[5] for s in seen) and any(s[1] > 0.0 for s in seen)
    ck = os.path.dirname(out_dir)
    emb0 = load_file(os.path.join(ck, "checkpoint-0", "te_only_tinyxl_embeddings.safetensors"))
    emb1 = load_file(os.path.join(out_dir, "te_only_tinyxl_embeddings.safetensors"))
    assert all(torch.equal(emb0[k], emb1[k]) for k in emb0)                      # the token rows never moved
    a0, a1 = load_file(os.path.join(ck, "checkpoint-0", "te_only_tinyxl_lora.safetensors")), load_file(os.path.join(out_dir, "te_only_tinyxl_lora.safetensors"))
    te_keys = [k for k in a0 if k.startswith("lora_te") and not k.endswith("alpha")]
    assert te_keys, sorted(a0)[:8]
    assert sum(int(not torch.equal(a0[k], a1[k])) for k in te_keys) > len(te_keys) // 2      # ... the text-encoder adapters did
    ta = json.load(open(os.path.join(out_dir, "training_args.json")))
    assert all(np.isfinite(ta["training_attributes"]["losses"]["tot_loss"]))


@pytest.mark.parametrize("kw", [dict(aspect_ratio_bucketing=True)])
def test_unbuilt_fields_raise(tmp_path, monkeypatch, kw):
    monkeypatch.chdir(tmp_path)
    from sd_lora_trainer_amd import train as T
    cfg = TrainingConfig(lora_training_urls="synthetic:4", concept_mode="object", pretrained_model={"path": "synthetic:tiny15"}, seed=1, resolution=128,
                         train_batch_size=1, max_train_steps=3, **kw)
    with pytest.raises(NotImplementedError):
        next(T.train(cfg, runtime=unet_mod.Runtime("cpu", 1, act_dtype=torch.float32, ops=emu_ops)))


def test_tok_cond_reg_with_dora_text_encoder_adapters_trains(tmp_path, monkeypatch):
    """The last refused combination of optional switches (round 5: NotImplementedError): `tok_cond_reg_w` + `use_dora` + `text_encoder_lora_optimizer`
    (loss.py:207-211 over optimizer.py:157-202) - the regularisation captions' second pass through the weight-decomposed adapters."""
    monkeypatch.chdir(tmp_path)
    from sd_lora_trainer_amd import train as T
    cfg = TrainingConfig(lora_training_urls="synthetic:4", concept_mode="object", pretrained_model={"path": "synthetic:tiny15"}, seed=1, resolution=128,
                         train_batch_size=1, max_train_steps=3, tok_cond_reg_w=0.1, text_encoder_lora_optimizer="adamw", use_dora=True,
                         checkpointing_steps=1000, validation_img_size=[128, 128], n_sample_imgs=1)
    progress, (config, out_dir) = _run(T.train(cfg, runtime=unet_mod.Runtime("cpu", 1, act_dtype=torch.float32, ops=emu_ops)))
    assert progress[-1] == 1.0
    ta = json.load(open(os.path.join(out_dir, "training_args.json")))
    assert all(np.isfinite(ta["training_attributes"]["losses"]["tot_loss"]))


def test_real_unet_without_text_encoder_weights_is_an_error(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    from safetensors.torch import save_file
    from sd_lora_trainer_amd import topology
    from sd_lora_trainer_amd import train as T
    sd = {k: torch.zeros(s) for k, s in topology.param_shapes(topology.CONFIGS["tiny15"]).items()}
    save_file(sd, str(tmp_path / "unet.safetensors"))
    topology.CONFIGS["sd15_test_alias"] = topology.CONFIGS["tiny15"]
    cfg = TrainingConfig(lora_training_urls="synthetic:4", concept_mode="object", pretrained_model={"path": str(tmp_path / "unet.safetensors")}, seed=1, resolution=128,
                         train_batch_size=1, max_train_steps=3)
    monkeypatch.setitem(topology.CONFIGS, "sd15", topology.CONFIGS["tiny15"])
    with pytest.raises(ValueError, match="text_encoder_path"):
        next(T.train(cfg, runtime=unet_mod.Runtime("cpu", 1, act_dtype=torch.float32, ops=emu_ops)))


@pytest.mark.parametrize("version", ["tiny15", "tinyxl"])
def test_train_with_dora(tmp_path, monkeypatch, version):
    """use_dora=True (optimizer.py:86-95; the hyper-parameter sweep's variant, create_hyperparam_sweep.py:77): the L1 penalty and the
    weight decay are switched off (config.py:153-157), the magnitudes are trained and leave as `dora_scale` next to down / up / alpha."""
    monkeypatch.chdir(tmp_path)
    from safetensors.torch import load_file
    from sd_lora_trainer_amd import train as T
    from sd_lora_trainer_amd.checkpoint import kohya_to_lora
    cfg = TrainingConfig(lora_training_urls="synthetic:4", concept_mode="object", pretrained_model={"path": f"synthetic:{version}"}, seed=1,
                         resolution=256 if version == "tinyxl" else 128, train_batch_size=2, max_train_steps=6, lora_rank=4, unet_lr=2e-3, ti_lr=1e-3, use_dora=True)
    assert cfg.l1_penalty == 0.0 and cfg.lora_weight_decay == 0.0
    rt = unet_mod.Runtime("cpu", 2, act_dtype=torch.float32, ops=emu_ops)
    _, (config, out_dir) = _run(T.train(cfg, runtime=rt))
    name = [n for n in os.listdir(out_dir) if n.endswith("_lora.safetensors")][0]
    sd = load_file(os.path.join(out_dir, name))
    lora = kohya_to_lora(sd)
    assert lora and all(len(v) == 3 for v in lora.values())
    conv = [k for k in lora if k.endswith("conv2")]
    lin = [k for k in lora if k.endswith("to_q")]
    assert lora[conv[0]][2].dim() == 4 and lora[conv[0]][2].shape[1] == lora[conv[0]][1].shape[0]      # [1, Cout, 1, 1]
    assert lora[lin[0]][2].shape == (lora[lin[0]][1].shape[0],)
    assert all(float(v[2].min()) > 0 for v in lora.values())
    assert json.load(open(os.path.join(out_dir, "adapter_config.json")))["use_dora"] is True
    e0 = kohya_to_lora(load_file(os.path.join(os.path.dirname(out_dir), "checkpoint-0", name))) if os.path.basename(out_dir) != "checkpoint-0" else None
    if e0 is not None:
        assert any(not torch.equal(e0[k][2], lora[k][2]) for k in lora)       # the magnitudes moved
