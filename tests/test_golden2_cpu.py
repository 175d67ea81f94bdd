"""Host mirrors against the round-2 fixtures generated from the reference's own code (oracle/gen_golden2.py):
prompt handling (inference.py:40-127, utils.py:27-47), learning-rate tables (main.py:236-240, 265-291 - SURVEY 8c viii),
token initialisation statistics (embedding_handler.py:157-223 - SURVEY 8c x)."""
import json
import os
import types

import pytest
import torch

from sd_lora_trainer_amd import prompts, schedule


def test_prompt_helpers_match_reference(golden_dir):
    g = json.load(open(os.path.join(golden_dir, "prompts.json")))
    for c in g["fix"]:
        assert prompts.fix_prompt(c["inp"]) == c["out"], c
    for c in g["replace"]:
        assert prompts.replace_in_string(c["s"], c["r"]) == c["out"], c
    assert prompts.NEGATIVE_PROMPT == g["negative_prompt"]
    assert len(g["prepare"]) >= 150
    for c in g["prepare"]:
        got = prompts.prepare_prompt_for_lora(c["prompt"], {"TOK": "<s0><s1><s2>"}, c["trigger_text"], c["name"], c["mode"], interpolation=c["interpolation"])
        assert got == c["out"], (c, got)


def test_lr_tables_match_reference(golden_dir):
    tables = json.load(open(os.path.join(golden_dir, "lr_schedule.json")))
    assert len(tables) >= 6
    for t in tables:
        cfg = types.SimpleNamespace(**t["config"])
        assert schedule.base_unet_lr(cfg.is_lora, cfg.disable_ti) == t["base_unet_lr"]
        for (epoch, step, gs, cf, lr_ti, lr_te, lr_unet) in t["rows"]:
            f = schedule.completion_fraction(epoch, step, t["steps_per_epoch"], cfg.num_train_epochs)
            assert f == pytest.approx(cf, rel=1e-12, abs=1e-15)
            lrs = schedule.learning_rates(cfg, gs, f, ti_active=lr_ti is not None, text_lora_active=t["text_lora"])
            assert lrs["unet"] == pytest.approx(lr_unet, rel=1e-12, abs=0)
            if lr_ti is not None and cfg.ti_optimizer != "prodigy":
                assert lrs["textual_inversion"] == pytest.approx(lr_ti, rel=1e-12, abs=0)
            if cfg.ti_optimizer == "prodigy":
                assert lr_ti == -1.0 and lrs["textual_inversion"] == 1.0        # the loop leaves a Prodigy group's lr untouched (main.py:269)
            if t["text_lora"]:
                assert lrs["text_encoders"] == pytest.approx(lr_te, rel=1e-12, abs=0)


def test_token_init_statistics_match_reference(golden_dir):
    """TokenEmbeddingsHandler.initialize_new_tokens on the fixture's tables: same train ids / no-update index; the std target is
    the mean per-row std of the RESIZED table; the drawn rows have exactly that mean per-row std (the reference's property,
    `row_std_mean == std_token_embedding` in the fixture)."""
    import sd_lora_trainer_amd.unet as M
    from sd_lora_trainer_amd.embedding_handler import TokenEmbeddingsHandler
    from sd_lora_trainer_amd.ti import TiState
    from tests import emu_ops
    recs = torch.load(os.path.join(golden_dir, "token_init.pt"))
    for rec in recs:
        rt = M.Runtime("cpu", 1, act_dtype=torch.float32, ops=emu_ops)
        encs = []
        for i in range(2):
            assert float(rec[f"row_std_mean_{i}"]) == pytest.approx(float(rec[f"std_token_embedding_{i}"]), rel=1e-5)
            tab = rec[f"table_{i}"].clone()
            # the reference's table right after `resize_token_embeddings`: pretrained rows + 3 new rows; stand-in for the new rows'
            # (version-dependent) resize initialisation: N(0, 0.02)
            tab[-3:] = 0.02 * torch.randn(3, tab.shape[1], generator=torch.Generator().manual_seed(5))
            encs.append(types.SimpleNamespace(table=tab, V=tab.shape[0], D=tab.shape[1]))
        ti = TiState(rt, encs, 3)
        h = TokenEmbeddingsHandler(ti, ["<s0>", "<s1>", "<s2>"])
        assert h.train_ids == rec["train_ids"]
        rows = h.initialize_new_tokens(seed=rec["seed"])
        for i, r in enumerate(rows):
            target = float(h.embeddings_settings[f"std_token_embedding_{i}"])
            # 3 stand-in rows of 689 move the target by < 0.5 %
            assert target == pytest.approx(float(rec[f"std_token_embedding_{i}"]), rel=5e-3)
            assert float(r.std(dim=1).mean()) == pytest.approx(target, rel=1e-5)
            assert torch.equal(h.embeddings_settings[f"index_no_updates_{i}"], rec[f"index_no_updates_{i}"])
            assert torch.allclose(encs[i].table[-3:], r)                       # the tables the encoders gather from hold the new rows
            assert torch.equal(encs[i].table[:-3], rec["pretrained"][i])      # every other row untouched
        # the regulariser's statistics are taken over the whole table after the initialisation (loss.py:190-193, 263-265)
        for i, (tm, tv) in enumerate(ti.stats):
            stds = encs[i].table.float().std(-1)
            assert tm == pytest.approx(float(stds.mean()), rel=1e-6) and tv == pytest.approx(float(stds.std() ** 2 / stds.mean()), rel=1e-6)
