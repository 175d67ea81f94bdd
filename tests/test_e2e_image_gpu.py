"""north_star's second parity clause - "per-step MSE loss AND FINAL SAMPLED IMAGES match the reference CPU/diffusers path on fixed seeds within
stated fp tolerance" - end to end, on one set of weights held by both sides:

    8 LoRA + textual-inversion training steps (main.py:263-382; injected latents / noise / timesteps / captions)
 -> the reference's render settings (trainer/inference.py:289-406, main.py:433-447): with- and without-concept conditionings from each side's OWN
    trained token rows, blend_conditions at sample_imgs_lora_scale, 25 Euler-"trailing" steps, classifier-free guidance 8, adapters at the
    render scale
 -> vae.decode(latents / scaling_factor) -> [0, 1] -> uint8 image

HIP side: step.TrainStep (hipGraph replays from step 1), clip text towers, sampler.LatentSampler on an inference UNet (batch 2), vae.VaeDecoder.
Oracle side: oracle/step_ref.RefTrainer + oracle/sampler_ref.sample_latents + oracle/vae_ref.decode (fp32, CPU).

Stated tolerances (bf16 storage on the HIP side; guidance 8 amplifies the difference of two bf16 forwards at each of the 25 steps, and the
training that precedes it hands the two samplers adapters that already differ by the bf16 noise of 8 AdamW steps): per-step image loss 1 %
relative; final latents cosine >= 0.9995 and relative L2 <= 4 %; decoded uint8 image PSNR >= 40 dB, mean absolute difference <= 2 levels and
maximum <= 24 levels of 255 - the bars sit at two to three times the error measured on the MI355X (recorded per case in gpurun_out/parity_report_e2e.json,
committed as profiles/r04_parity_report_e2e.json)."""
import json
import math
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

# measured (profiles/r04_parity_report_e2e.json): tinyxl latents cos 0.99985 / rel 1.7 %, image PSNR 45.0 dB, mean |diff| 1.0 level, max 8 levels;
# real SDXL: per-step losses within 0.2 %, latents cos 0.99990 / rel 1.4 %, PSNR 45.5 dB, mean 0.94, max 8 levels (every pixel within 8 of 255)
TOL = {"tinyxl": dict(loss=1e-2, lat_cos=0.9995, lat_rel=0.04, psnr=40.0, mean_abs=2.0, max_abs=24.0),
       "sdxl": dict(loss=1e-2, lat_cos=0.9995, lat_rel=0.04, psnr=40.0, mean_abs=2.0, max_abs=24.0),
       "sd15": dict(loss=1e-2, lat_cos=0.9995, lat_rel=0.04, psnr=40.0, mean_abs=2.0, max_abs=24.0)}
REPORT = {}


def _psnr(a, b):
    mse = float(((a.double() - b.double()) ** 2).mean())
    return 99.0 if mse == 0 else 10.0 * math.log10(255.0 ** 2 / mse)


def _prompt_ids(vc, words, with_tokens):
    l = [vc.bos] + words[:3] + (vc.train_ids if with_tokens else []) + words[3:] + [vc.eos]
    ids = torch.full((1, 77), vc.eos, dtype=torch.int64)
    ids[0, :len(l)] = torch.tensor(l)
    return ids


def run_e2e(version, kinds, vae_kind, h, rank, *, n_train=8, n_render=25, lora_scale=0.75, guidance=8.0, B=1):
    from oracle import sampler_ref as SR
    from oracle import step_ref as R
    from oracle import unet_ref as U
    from oracle import vae_ref as V
    import sd_lora_trainer_amd.clip as CL
    import sd_lora_trainer_amd.step as S
    import sd_lora_trainer_amd.unet as M
    from sd_lora_trainer_amd import sampler as SM
    from sd_lora_trainer_amd import topology, vae
    from tests.test_real_topology_gpu import NTOK, Vocab, _batch, _bf16_exact, _build_product, _cos_rel, _hf_clip, _set, _unet_state
    cfg = U.CONFIGS[version]
    xl = bool(cfg["addition"])
    real = not version.startswith("tiny")
    sd = _unet_state(version) if real else _bf16_exact(U.init_unet_state(cfg, seed=0))
    vcfg = V.CONFIGS[vae_kind]
    vsd = _bf16_exact(V.init_state(vcfg, seed=0))
    vc = Vocab(topology.CLIP_CONFIGS[kinds[0]]["vocab"])
    lora = U.init_lora(cfg, rank, seed=1, b_std=0.02)
    hf = [_hf_clip(k, 11 + i) for i, k in enumerate(kinds)]
    w_ta, lr, lr_ti = 2e-2, 4e-4, 1e-3
    rt, unet, ts = _build_product(version, B, h, sd, lora, hf, rank, kinds=kinds, snr_gamma=5.0, l1_penalty=0.03, weight_decay=0.004,
                                  token_attention_loss_w=w_ta, ti_std_loss_w=0.01)
    ref = R.RefTrainer(cfg, sd, lora, text_models=hf, n_tokens=NTOK, train_ids=vc.train_ids, snr_gamma=5.0, l1_penalty=0.03, weight_decay=0.004,
                       token_attention_loss_w=w_ta, ti_std_loss_w=0.01)
    tol = TOL[version]
    # ---- 1. training: 8 steps on two alternating batches, per-step image loss against the oracle's
    batches = [_batch(cfg, B, h, 3, [10, 900, 500, 999], vc), _batch(cfg, B, h, 4, [700, 50, 300, 850], vc)]
    losses = []
    for step in range(n_train):
        b = batches[step % 2]
        tid = _set(ts, b, xl, h, len(hf))
        o = ref.step(b["latent"], b["noise"], b["t"], b["mask"], lr=lr, lr_ti=lr_ti, ids=b["ids"], caption_token_lists=b["lists"], time_ids=tid)
        if step == 1:
            ts.capture(warmup=1)
        ts.run(lr, lr_ti=lr_ti)
        torch.cuda.synchronize()
        losses.append((float(ts.loss), o["img_loss"]))
    for i, (l, lo) in enumerate(losses):
        assert abs(l - lo) <= tol["loss"] * abs(lo), f"step {i}: image loss {l} vs oracle {lo}; {losses}"
    # ---- 2. conditionings from each side's own trained token rows: with the concept, without it, and the negative prompt
    g = torch.Generator().manual_seed(21)
    words = torch.randint(3, vc.bos - 1, (6,), generator=g).tolist()
    neg_words = torch.randint(3, vc.bos - 1, (4,), generator=g).tolist()
    ids_lora, ids_zero, ids_neg = _prompt_ids(vc, words, True), _prompt_ids(vc, words, False), _prompt_ids(vc, neg_words, False)
    tid2 = torch.tensor([[8. * h, 8. * h, 0, 0, 8. * h, 8. * h]])

    def oracle_embeds(ids):
        with torch.no_grad():
            c, add = ref.conditioning(torch.cat([ids_neg, ids], 0), tid2.repeat(2, 1) if xl else None)
        return (c[1:2], c[0:1]) + ((add["text_embeds"][1:2], add["text_embeds"][0:1]) if xl else ())

    rt2 = M.Runtime("cuda:0", 2)
    encs = []
    for i, (m, kd) in enumerate(zip(hf, kinds)):
        c = topology.CLIP_CONFIGS[kd]
        csd = {k: v.detach().clone() for k, v in m.state_dict().items()}           # the oracle trained these tables in place: the inference encoders
        enc = CL.ClipTextEncoder(rt2, f"rte{i + 1}", csd, heads=c["heads"], act=c["act"], mode="penultimate" if xl else "last",
                                 with_projection=bool(c["proj"]), n_train=NTOK)
        enc.table[enc.V - NTOK:].copy_(ts.text.encoders[i].table[enc.V - NTOK:])   # ... take the rows the HIP side trained (train.Renderer.sync)
        encs.append(enc)
    text2 = S.TextStack(rt2, encs, pool_mode="argmax" if xl else "first_eos", eos_token_id=vc.eos)
    ctx2 = rt2.zeros(2 * M.CTX_PAD, cfg["cross_dim"])

    def product_embeds(ids):
        text2.set_ids([torch.cat([ids_neg, ids], 0).cuda()] * len(encs))
        pooled = text2.forward(ctx2)
        cv = ctx2.view(2, M.CTX_PAD, -1)[:, :77].float().clone()
        return (cv[1:2], cv[0:1]) + ((pooled.float()[1:2].clone(), pooled.float()[0:1].clone()) if pooled is not None else ())

    emb_o, _ = SM.blend_conditions(oracle_embeds(ids_zero), oracle_embeds(ids_lora), lora_scale)          # (blend_conditions is pinned by tests/golden on its own)
    with torch.no_grad():
        emb_p, _ = SM.blend_conditions(product_embeds(ids_zero), product_embeds(ids_lora), lora_scale)
    c_cos, c_rel = _cos_rel(emb_p[0], emb_o[0])
    # ---- 3. 25 trailing Euler steps at guidance 8 with the trained adapters at the render scale
    noise = torch.randn(1, 4, h, h, generator=g)
    lora_o = {k: tuple(t.detach() for t in v) for k, v in ref.lora.items()}
    lat_o = SR.sample_latents(cfg, sd, lora_o, lora_scale, emb_o, noise, n_render, guidance_scale=guidance)
    unet2 = M.UNet(rt2, topology.CONFIGS[version], sd, lora_rank=rank)
    unet2.arena.params.copy_(unet.arena.params)
    unet2.arena.refresh_shadows()
    smp = SM.LatentSampler(rt2, unet2)
    smp.set_lora_scale(lora_scale)
    lat_p = smp.sample(tuple(None if e is None else e.cuda() for e in emb_p), h, h, steps=n_render, guidance_scale=guidance, latents=noise.cuda()).cpu()
    assert torch.isfinite(lat_p).all()
    l_cos, l_rel = _cos_rel(lat_p, lat_o)
    # ---- 4. VAE decode -> uint8 image
    img_o = (V.postprocess(V.decode(vcfg, vsd, lat_o / cfg["scaling_factor"]))[0].permute(1, 2, 0) * 255).round().clamp(0, 255)
    dec = vae.VaeDecoder(M.Runtime("cuda:0", 1), vsd)
    img_p = (vae.postprocess(dec.decode(lat_p.cuda() / cfg["scaling_factor"]))[0].permute(1, 2, 0).float().cpu() * 255).round().clamp(0, 255)
    d = (img_p - img_o).abs()
    rep = dict(train_losses=losses, conditioning=dict(cos=c_cos, rel=c_rel), latents=dict(cos=l_cos, rel=l_rel, steps=n_render, guidance=guidance, lora_scale=lora_scale),
               image=dict(psnr_db=_psnr(img_p, img_o), mean_abs_levels=float(d.mean()), max_abs_levels=float(d.max()), frac_within_8=float((d <= 8).float().mean()),
                          shape=list(img_o.shape), oracle_std_levels=float(img_o.std())))
    REPORT[f"{version}-h{h}"] = rep
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, "parity_report_e2e.json"), "w") as fh:
            json.dump(REPORT, fh, indent=1)
    assert l_cos >= tol["lat_cos"] and l_rel <= tol["lat_rel"], f"final latents after {n_render} steps: cos {l_cos} rel {l_rel}; {rep}"
    assert rep["image"]["psnr_db"] >= tol["psnr"] and rep["image"]["mean_abs_levels"] <= tol["mean_abs"] and rep["image"]["max_abs_levels"] <= tol["max_abs"], rep
    assert rep["image"]["oracle_std_levels"] > 1.0, "the oracle's image is constant: the comparison says nothing"
    return rep


def test_train_render_decode_tinyxl():
    """toy SDXL-shaped topology (two text towers, add-embedding), 32 x 32 latent (64 tokens at the coarsest level: the smallest the token-attention GEMMs take), the toy autoencoder"""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    run_e2e("tinyxl", ["tiny_l", "tiny_g"], "tiny", 32, 8)


def test_train_render_decode_sdxl_real_topology():
    """the REAL SDXL UNet / CLIP-L / OpenCLIP-bigG / AutoencoderKL topologies (random-init, bf16-exact weights) at a 32 x 32 latent = 256 px image:
    8 training steps, the full 25 render steps"""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    run_e2e("sdxl", ["clip_l", "clip_g"], "sd", 32, 16)


@pytest.mark.skipif(os.environ.get("SDLT_E2E_1024") != "1", reason="the metric's own size (128 x 128 latent = 1024 px): ~60 fp32 oracle UNet passes at 13.7 TFLOP each take ~10 minutes "
                    "of host time - run as a report (SDLT_E2E_1024=1 python -m pytest tests/test_e2e_image_gpu.py -k 1024px; profiles/r06_parity_report_e2e_1024.json), not in the driver's suite")
def test_train_render_decode_sdxl_real_topology_1024px():
    """north_star's image clause AT THE METRIC'S SIZE: the real SDXL / CLIP / AutoencoderKL topologies, 128 x 128 latent = a 1024 x 1024 image:
    4 training steps, the full 25 trailing Euler steps at guidance 8, VAE decode, uint8 image against the oracle pipeline - same bars as the 256 px case."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    run_e2e("sdxl", ["clip_l", "clip_g"], "sd", 128, 16, n_train=4)


def test_train_render_decode_sd15_real_topology():
    """the REAL SD1.5 UNet / CLIP-L / AutoencoderKL topologies (cfg2's model; head widths 40 / 80 / 160, one text tower, `first_eos` pooling, no
    add-embedding) at a 32 x 32 latent, batch 2 in training"""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    run_e2e("sd15", ["clip_l"], "sd", 32, 16, B=2)
