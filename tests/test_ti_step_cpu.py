"""Host-logic test (CPU) of the full textual-inversion step: text encoders (trainable token rows) -> UNet+LoRA ->
diffusion loss + token-attention (DAAM) loss + token-std regulariser -> gradients of the TI rows and of every LoRA
tensor, against an fp32 oracle composed of Hugging Face CLIP (transformers), oracle/unet_ref.py and oracle/loss_ref.py."""
import pytest
import torch

from oracle import loss_ref as L
from oracle import unet_ref as U
from tests import emu_ops

import sd_lora_trainer_amd.clip as clip_mod
import sd_lora_trainer_amd.step as step_mod
import sd_lora_trainer_amd.unet as unet_mod
from sd_lora_trainer_amd import topology

transformers = pytest.importorskip("transformers")

V, NTOK, EOS, BOS = 203, 3, 199, 198
TRAIN_IDS = [200, 201, 202]


def _hf(act, with_proj, hidden, heads, seed, proj=64):
    from transformers import CLIPTextConfig, CLIPTextModel, CLIPTextModelWithProjection
    torch.manual_seed(seed)
    cfg = CLIPTextConfig(vocab_size=V, hidden_size=hidden, intermediate_size=2 * hidden, num_hidden_layers=3, num_attention_heads=heads,
                         max_position_embeddings=77, hidden_act=act, projection_dim=proj, eos_token_id=EOS, bos_token_id=BOS, pad_token_id=EOS)
    m = (CLIPTextModelWithProjection if with_proj else CLIPTextModel)(cfg).eval()
    for p in m.parameters():
        if p.dim() == 1:
            p.data.add_(0.05 * torch.randn_like(p))
    return m


def _captions(B):
    lists = [[BOS, 5, 17, 33] + TRAIN_IDS + [41, EOS], [BOS, 7, 9, EOS]][:B]     # 2nd caption lacks the TI tokens
    ids = torch.full((B, 77), EOS, dtype=torch.int64)
    for b, l in enumerate(lists):
        ids[b, :len(l)] = torch.tensor(l)
    return lists, ids


@pytest.mark.parametrize("version,B", [("tiny15", 2), ("tinyxl", 1), ("tinyxl", 2)])
def test_ti_step_matches_oracle(version, B):
    cfg = U.CONFIGS[version]
    xl = cfg["addition"]
    rank, h, w_ta, w_std = 4, 16, 2e-2, 0.01
    sd = U.init_unet_state(cfg, seed=0)
    lora = U.init_lora(cfg, rank, seed=1, b_std=0.05)
    if xl:
        hf = [_hf("quick_gelu", False, 64, 1, 11), _hf("gelu", True, 64, 1, 12, proj=cfg["proj_class_in"] - 6 * cfg["addition_time_embed_dim"])]
    else:
        hf = [_hf("quick_gelu", False, 64, 2, 11)]
    g = torch.Generator().manual_seed(3)
    latent = torch.randn(B, 4, h, h, generator=g) * cfg["scaling_factor"]
    noise = torch.randn(B, 4, h, h, generator=g)
    mask = (torch.rand(B, 1, h, h, generator=g) * 0.95 + 0.05).repeat(1, 4, 1, 1).contiguous()
    t = torch.tensor([10, 900][:B])
    tid = torch.tensor([[1024., 1024, 0, 0, 128, 128]] * B) if xl else None
    lists, ids = _captions(B)

    # ------------------------------------------------------------------ oracle (autograd)
    embs = [m.get_input_embeddings().weight for m in hf]
    outs = [m(input_ids=ids, output_hidden_states=True) for m in hf]
    if xl:
        ctx = torch.cat([outs[0].hidden_states[-2], outs[1].hidden_states[-2]], dim=-1)
        add = {"text_embeds": outs[1].text_embeds, "time_ids": tid}
    else:
        ctx, add = outs[0].last_hidden_state, None
    lora_g, params = {}, []
    for k, (A, Bm) in lora.items():
        A, Bm = A.clone().requires_grad_(True), Bm.clone().requires_grad_(True)
        lora_g[k] = (A, Bm)
        params += [A, Bm]
    acp = L.ddpm_alphas_cumprod()
    noisy = L.add_noise(acp, latent, noise, t)
    pred, daam = U.unet_forward(cfg, sd, noisy, t, ctx, add, lora=lora_g, return_daam=True)
    img_loss = L.diffusion_loss(pred, noise, noisy, mask, acp, t, snr_gamma=5.0)
    ta = L.token_attention_loss(L.daam_stack([s for _, s in daam], 1.0), mask, lists, TRAIN_IDS)
    reg = torch.stack([L.DistributionStats(e.detach()[:-NTOK]).std_loss(e[-NTOK:]) for e in embs]).mean()
    loss = img_loss + w_ta * ta + w_std * reg
    grads = torch.autograd.grad(loss, params + embs)
    g_lora = {k: (grads[2 * i], grads[2 * i + 1]) for i, k in enumerate(lora)}
    g_rows = [ge[-NTOK:] for ge in grads[len(params):]]

    # ------------------------------------------------------------------ the plan, through the op emulation
    rt = unet_mod.Runtime("cpu", B, act_dtype=torch.float32, ops=emu_ops)
    unet = unet_mod.UNet(rt, topology.CONFIGS[version], sd, lora_rank=rank)
    unet.arena.load(lora)
    sds = [{k: v.detach() for k, v in m.state_dict().items()} for m in hf]
    if xl:
        encs = [clip_mod.ClipTextEncoder(rt, "te1", sds[0], heads=1, act="quick_gelu", mode="penultimate", with_projection=False, n_train=NTOK),
                clip_mod.ClipTextEncoder(rt, "te2", sds[1], heads=1, act="gelu", mode="penultimate", with_projection=True, n_train=NTOK)]
    else:
        encs = [clip_mod.ClipTextEncoder(rt, "te1", sds[0], heads=2, act="quick_gelu", mode="last", with_projection=False, n_train=NTOK)]
    text = step_mod.TextStack(rt, encs, pool_mode="first_eos", eos_token_id=EOS)
    ts = step_mod.TrainStep(rt, unet, latent_hw=(h, h), snr_gamma=5.0, l1_penalty=0.0, weight_decay=0.0, text=text, n_tokens=NTOK,
                            token_attention_loss_w=w_ta, ti_std_loss_w=w_std)
    ts.set_batch(latent, noise, t, mask, time_ids=tid, ids=[ids] * len(encs), caption_token_lists=lists)
    ts.forward_backward()
    torch.testing.assert_close(ts.loss[0], img_loss.detach(), rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(ts.ta.loss[0], ta.detach(), rtol=1e-3, atol=1e-5)
    torch.testing.assert_close(ts.ti.reg_loss[0], (w_std * reg).detach(), rtol=1e-3, atol=1e-7)
    for got, ref in zip(ts.ti.grad_rows, g_rows):
        scale = float(ref.abs().max())
        assert scale > 0 and float((got - ref).abs().max()) <= 3e-3 * scale, (float((got - ref).abs().max()), scale)
    got = unet.arena.export("grads")
    for k in g_lora:
        for a, b_ in zip(got[k], g_lora[k]):
            scale = max(float(b_.abs().max()), 1e-8)
            assert float((a - b_).abs().max()) <= 3e-3 * scale + 1e-7, (k, float((a - b_).abs().max()), scale)

    # one TI optimiser step: rows-only AdamW == torch.optim.AdamW on the rows
    rows0 = [r.clone() for r in ts.ti.rows]
    grows = [r.clone() for r in ts.ti.grad_rows]
    ts.set_hyper(1e-3, lr_ti=1e-3)
    ts.optimizer_step()
    for r0, gr, r1, enc in zip(rows0, grows, ts.ti.rows, encs):
        p, m, v = r0.clone(), torch.zeros_like(r0), torch.zeros_like(r0)
        L.adamw_step(p, gr, m, v, 1, 1e-3, weight_decay=0.0)
        torch.testing.assert_close(r1, p, rtol=1e-5, atol=1e-7)
        torch.testing.assert_close(enc.table[-NTOK:].float(), p, rtol=1e-5, atol=1e-7)    # gathered table was refreshed
