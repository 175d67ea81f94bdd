"""Host-logic test (CPU): the CLIP text-encoder plan (sd-lora-trainer_amd/clip.py) through the torch emulation of the
ops against Hugging Face `CLIPTextModel(WithProjection)` (transformers is the third-party package the reference's
encode_prompt runs; random-init configs, no checkpoint needed): hidden states, pooled output and the gradient of the
trainable token rows."""
import pytest
import torch

from tests import emu_ops

import sd_lora_trainer_amd.clip as clip_mod
import sd_lora_trainer_amd.unet as unet_mod

transformers = pytest.importorskip("transformers")


def _hf(act, with_proj, seed):
    from transformers import CLIPTextConfig, CLIPTextModel, CLIPTextModelWithProjection
    torch.manual_seed(seed)
    cfg = CLIPTextConfig(vocab_size=203, hidden_size=128, intermediate_size=256, num_hidden_layers=3, num_attention_heads=2,
                         max_position_embeddings=77, hidden_act=act, projection_dim=64, eos_token_id=199, bos_token_id=198, pad_token_id=199)
    m = (CLIPTextModelWithProjection if with_proj else CLIPTextModel)(cfg).eval()
    for p in m.parameters():     # make biases / LN params non-trivial
        if p.dim() == 1:
            p.data.add_(0.05 * torch.randn_like(p))
    return m, cfg


@pytest.mark.parametrize("act,mode,with_proj", [("quick_gelu", "last", False), ("quick_gelu", "penultimate", False), ("gelu", "penultimate", True)])
def test_clip_encoder_matches_transformers(act, mode, with_proj):
    m, cfg = _hf(act, with_proj, seed=5)
    B, n_train = 2, 3
    g = torch.Generator().manual_seed(1)
    ids = torch.randint(0, 190, (B, 77), generator=g)
    ids[:, 0] = 198
    ids[0, 3:6] = torch.tensor([200, 201, 202])       # the trainable tokens (last rows of the table)
    ids[1, 10] = 201
    ids[0, 20:] = 199
    ids[1, 30:] = 199
    emb = m.get_input_embeddings().weight
    out = m(input_ids=ids, output_hidden_states=True)
    hidden_ref = out.hidden_states[-2] if mode == "penultimate" else out.last_hidden_state
    gh = torch.randn(hidden_ref.shape, generator=g)
    loss = (hidden_ref * gh).sum()
    gp = None
    if with_proj:
        gp = torch.randn(out.text_embeds.shape, generator=g)
        loss = loss + (out.text_embeds * gp).sum()
    (gemb,) = torch.autograd.grad(loss, emb)

    rt = unet_mod.Runtime("cpu", B, act_dtype=torch.float32, ops=emu_ops)
    sd = {k: v.detach() for k, v in m.state_dict().items()}
    enc = clip_mod.ClipTextEncoder(rt, "te", sd, heads=2, act=act, mode=mode, with_projection=with_proj, n_train=n_train)
    pool_rows = torch.arange(B) * clip_mod.TP + (ids == cfg.eos_token_id).int().argmax(-1)
    ctx = torch.zeros(B * clip_mod.TP, 128 + 64)
    hidden, pooled = enc.forward(ids, B, hidden_out=ctx[:, 64:], pool_rows=pool_rows)
    got = ctx[:, 64:].reshape(B, clip_mod.TP, 128)[:, :77]
    torch.testing.assert_close(got, hidden_ref, rtol=1e-4, atol=1e-4)
    if with_proj:
        torch.testing.assert_close(pooled, out.text_embeds, rtol=1e-4, atol=1e-4)
    dctx = torch.zeros(B * clip_mod.TP, 128 + 64)
    dctx[:, 64:].reshape(B, clip_mod.TP, 128)[:, :77] = gh
    grad_rows = torch.full((n_train, 128), 7.0)
    enc.backward(dctx[:, 64:], gp, grad_rows)
    torch.testing.assert_close(grad_rows, gemb[-n_train:], rtol=1e-3, atol=1e-4)
    assert float(gemb[-n_train:].abs().max()) > 1e-3
