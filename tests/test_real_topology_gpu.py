"""Parity on the REAL SDXL / SD1.5 topologies (BASELINE.json configs 2, 3, 5), not on toy models: the exact
UNet2DConditionModel wiring and widths (SDXL: 2,567,463,684 parameters, 10-deep 1280-wide transformer stacks, 20 heads;
SD1.5: 859,520,964), the exact CLIP-L / OpenCLIP-bigG text towers (12 x 768, 32 x 1280 + projection), rank-16 adapters on all
560+17 / 128+22 target layers, random-init weights (no checkpoints offline), bf16-exact so both sides see the same model.

  (a) ONE LoRA + textual-inversion step at a reduced latent (32 x 32; the CPU fp32 oracle takes seconds) against
      oracle/step_ref.py (main.py:263-382): prediction, image loss, token-attention loss, LoRA gradient, token-row gradients;
  (b) a 6-step LOSS TRAJECTORY under AdamW (both optimizers live, L1 penalty, regulariser) with injected latents / noise /
      timesteps / captions: every step's losses against the oracle's, and the final LoRA / token-row state; the same with DoRA
      adapters (use_dora, incl. the magnitude gradients on their own), and the first step at the sweep's other ranks (24, 64);
  (c) at the FULL BASELINE size (SDXL 128 x 128 batch 1; SD1.5 64 x 64 batch 4): ONE whole step against the fp32 oracle (prediction,
      losses, LoRA gradient, token-row gradients - the oracle step takes ~20 s on the box's 16 cores), and the size-independent properties:
      finite, hipGraph replay == eager gradients, optimizer state advances, loss goes down on a fixed batch;
  (d) the full fine-tune (cfg5) on the SDXL topology: EVERY parameter's gradient against oracle autograd.

Stated tolerances (bf16 storage of every activation, fp32 accumulation, ~200 GEMMs deep on SDXL):
  prediction max-abs <= 4e-2 of max|pred|; losses <= 2e-2 relative (token-attention loss 3e-2); LoRA-gradient cosine >= 0.99 and
  relative L2 <= 8e-2; token-row gradients cosine >= 0.985; trajectory: per-step image loss <= 3e-2 relative.
"""
import math
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

NTOK = 3


class Vocab:
    """Token ids of a CLIP vocabulary of `size` entries with the NTOK trigger tokens appended (embedding_handler.py:157-223)."""

    def __init__(self, size):
        self.size = size
        self.bos, self.eos = (49406, 49407) if size >= 49408 else (size - 2, size - 1)
        self.train_ids = [size + i for i in range(NTOK)]


REAL = Vocab(49408)


def _cos_rel(a, b):
    a, b = a.reshape(-1).double().cpu(), b.reshape(-1).double().cpu()
    return float(a @ b / (a.norm() * b.norm() + 1e-30)), float((a - b).norm() / (b.norm() + 1e-30))


def _bf16_exact(sd):
    return {k: v.to(torch.bfloat16).float() for k, v in sd.items()}


_CACHE = {}


def _unet_state(version):
    """bf16-exact random UNet weights of the real topology (fan-in scaled like oracle.unet_ref.init_unet_state), built once per
    version: drawn on the GPU when there is one (2.57 G normal draws take ~40 s on one host core) and handed to both sides."""
    from oracle import unet_ref as U
    if version not in _CACHE:
        _CACHE.clear()            # one 10 GB state at a time
        if torch.cuda.is_available():
            g = torch.Generator(device="cuda").manual_seed(0)
            sd = {}
            for n, shp in U.param_shapes(U.CONFIGS[version]).items():
                t = torch.randn(shp, generator=g, device="cuda", dtype=torch.float32)
                is_norm = (".norm" in n or n.startswith("conv_norm_out")) and len(shp) == 1
                t = t * (1.0 / math.sqrt(math.prod(shp[1:])) if len(shp) >= 2 else 0.02)
                if is_norm and n.endswith(".weight"):
                    t = 1.0 + t
                sd[n] = t.to(torch.bfloat16).float().cpu()
            _CACHE[version] = sd
        else:
            _CACHE[version] = _bf16_exact(U.init_unet_state(U.CONFIGS[version], seed=0))
    return _CACHE[version]


def _trained_like(sd, seed=17):
    """fan-in-scaled random weights give attention logits of unit spread - softmax near uniform over 1024 - 4096 keys, the q / k
    gradients numerically nothing (rms 1 % of the median adapter's), no outlier channels.  This is the same state pushed towards what a trained SDXL / SD1.5 looks
    like, for BOTH sides of the comparison: to_q and to_k of every attention doubled (logit spread x 4: a peaked softmax, q / k gradients that carry signal),
    1 % of the channels of every transformer's residual stream amplified 10 - 30 x (proj_in rows: the massive-activation channels of trained transformers),
    LayerNorm / GroupNorm gains 1 +- 0.3 with biases of 0.1.  bf16-exact like the state it starts from."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for n, t in sd.items():
        if n.endswith((".to_q.weight", ".to_k.weight")):
            t = t * 2.0
        elif n.endswith(".proj_in.weight") and t.dim() == 2:
            C = t.shape[0]
            idx = torch.randperm(C, generator=g)[: max(1, C // 100)]
            f = torch.ones(C)
            f[idx] = 10.0 + 20.0 * torch.rand(idx.numel(), generator=g)
            t = t * f[:, None]
        elif t.dim() == 1 and (".norm" in n or n.startswith("conv_norm_out")):
            r = torch.randn(t.shape, generator=g)
            t = (1.0 + 0.3 * r).clamp_min(0.2) if n.endswith(".weight") else 0.1 * r
        out[n] = t.to(torch.bfloat16).float() if t is not sd[n] else t
    return out


_HF = {}


def _hf_clip(kind, seed, legacy_eos=None):
    """Hugging Face CLIP text tower of the real size with the 3 new token rows appended (embedding_handler.py:157-223); built once
    per (kind, seed) - transformers' parameter initialisation of the 695 M-parameter bigG tower takes a minute - and copied per use
    (the oracle trains its copy's token table in place)."""
    import copy
    key = (kind, seed, legacy_eos)
    if key not in _HF:
        _HF[key] = _hf_clip_build(kind, seed, legacy_eos)
    return copy.deepcopy(_HF[key])


def _hf_clip_build(kind, seed, legacy_eos=None):
    from transformers import CLIPTextConfig, CLIPTextModel, CLIPTextModelWithProjection
    from sd_lora_trainer_amd import topology
    c = topology.CLIP_CONFIGS[kind]
    vc = Vocab(c["vocab"])
    legacy_eos = bool(c["proj"]) if legacy_eos is None else legacy_eos
    torch.manual_seed(seed)
    # SDXL's text_encoder_2 ships the legacy eos_token_id = 2 -> pooled output at argmax(input_ids) (transformers CLIPTextTransformer)
    cfg = CLIPTextConfig(vocab_size=c["vocab"] + NTOK, hidden_size=c["width"], intermediate_size=c["mlp"], num_hidden_layers=c["layers"],
                         num_attention_heads=c["heads"], max_position_embeddings=77, hidden_act=c["act"], projection_dim=c["proj"] or 768,
                         eos_token_id=2 if legacy_eos else vc.eos, bos_token_id=vc.bos, pad_token_id=1)
    m = (CLIPTextModelWithProjection if c["proj"] else CLIPTextModel)(cfg).eval()
    for p in m.parameters():
        if p.dim() == 1:
            p.data.add_(0.05 * torch.randn_like(p))
        p.data = p.data.to(torch.bfloat16).float()
    return m


def _captions(B, seed, vc=REAL):
    g = torch.Generator().manual_seed(seed)
    lists = []
    for b in range(B):
        words = torch.randint(3, vc.bos - 1, (7,), generator=g).tolist()
        if b % 4 == 3:
            lists.append([vc.bos] + words + [vc.eos])                 # a caption without the trigger tokens (loss.py:40-43)
        else:
            lists.append([vc.bos] + words[:3] + vc.train_ids + words[3:] + [vc.eos])
    ids = torch.full((B, 77), vc.eos, dtype=torch.int64)
    for b, l in enumerate(lists):
        ids[b, :len(l)] = torch.tensor(l)
    return lists, ids


def _batch(cfg, B, h, seed, tvals, vc=REAL):
    g = torch.Generator().manual_seed(seed)
    latent = torch.randn(B, 4, h, h, generator=g) * cfg["scaling_factor"]
    noise = torch.randn(B, 4, h, h, generator=g)
    mask = (torch.rand(B, 1, h, h, generator=g) * 0.95 + 0.05).repeat(1, 4, 1, 1).contiguous()
    t = torch.tensor(tvals[:B])
    lists, ids = _captions(B, seed, vc)
    return dict(latent=latent, noise=noise, mask=mask, t=t, lists=lists, ids=ids)


def _build_product(version, B, h, sd, lora, hf, rank, kinds=None, device="cuda:0", ops=None, act_dtype=torch.bfloat16, use_dora=False, **step_kw):
    import sd_lora_trainer_amd.clip as clip_mod
    import sd_lora_trainer_amd.step as step_mod
    import sd_lora_trainer_amd.unet as unet_mod
    from sd_lora_trainer_amd import topology
    xl = topology.CONFIGS[version]["addition"]
    rt = unet_mod.Runtime(device, B, act_dtype=act_dtype, ops=ops)
    unet = unet_mod.UNet(rt, topology.CONFIGS[version], sd, lora_rank=rank, use_dora=use_dora)
    unet.arena.load(lora)
    text = None
    if hf is not None:
        kinds = kinds or (["clip_l", "clip_g"] if xl else ["clip_l"])
        encs = []
        for i, (m, kd) in enumerate(zip(hf, kinds)):
            c = topology.CLIP_CONFIGS[kd]
            csd = {k: v.detach().clone() for k, v in m.state_dict().items()}      # (a CPU runtime would alias the oracle's tables)
            encs.append(clip_mod.ClipTextEncoder(rt, f"te{i + 1}", csd, heads=c["heads"], act=c["act"], mode="penultimate" if xl else "last",
                                                 with_projection=bool(c["proj"]), n_train=NTOK))
        text = step_mod.TextStack(rt, encs, pool_mode="argmax" if xl else "first_eos", eos_token_id=Vocab(topology.CLIP_CONFIGS[kinds[0]]["vocab"]).eos)
    ts = step_mod.TrainStep(rt, unet, latent_hw=(h, h), text=text, n_tokens=NTOK, **step_kw)
    return rt, unet, ts


def _set(ts, b, xl, h, n_enc):
    dev = ts.rt.device
    tid = torch.tensor([[1024., 1024, 0, 0, 8. * h, 8. * h]] * b["latent"].shape[0]) if xl else None
    ts.set_batch(b["latent"].to(dev), b["noise"].to(dev), b["t"].to(dev), b["mask"].to(dev), time_ids=tid.to(dev) if xl else None,
                 ids=[b["ids"]] * n_enc, caption_token_lists=b["lists"])
    return tid


# Two yardsticks (stated tolerances, bf16 storage / fp32 accumulation on the HIP side):
#   TOL_BF16      against the fp32 oracle (THE reference): what bf16 storage through ~40 layers leaves, plus any logic error
#   TOL_FAITHFUL  against the same oracle in its bf16-faithful mode (oracle/unet_ref.py: every tensor the HIP path stores in bf16 is rounded
#                 where it is stored, gradients included): rounding is (mostly) common to both sides, so the bars are 3-6x tighter and a
#                 wrong eps / bias / scale in ONE small adapter no longer hides in the noise of a 25 M-vector
# per-adapter: the gradient of EVERY adapter tensor on its own (577 / 150 adapted layers), worst adjusted relative error ceiling `ada_rel` (see _per_adapter)
# Measured on the real topologies (profiles/r03_parity_report.json): prediction 1.3-2.2 % of max-abs, flat LoRA gradient cos 0.99994-0.99998 /
# rel-L2 0.7-1.1 %, median adapter 0.6-1.1 %, worst adapter 1.4-10 % (the 10 %: self-attention q / k adapters whose gradient is 1 % of the
# median), token rows cos >= 0.9999 / rel-L2 0.6-1.3 %, image loss within 0.25 %.  Both oracle modes give the same figures to +-0.3 %: what
# separates the HIP path from the fp32 reference is NOT dominated by where activations are rounded, so the bars below sit 2-3x above the
# measurements for both (round 2: cos 0.99 / rel 8 % on the flat vector only).  LoRA displacement after 6 AdamW steps: cos 0.970 (SDXL: Adam turns the
# noise of the near-zero gradients into +-lr moves) / 0.9995 (SD1.5); floor 0.95 (round 2: 0.9).
# round 5, the two adapter classes on their own (profiles/r05_parity_report.json -> adapter_classes): normal-sized adapters (rms >= 0.1 median) worst plain relative error 1.4-5.4 %
# -> bar 10 %; the ~1 %-sized ones worst ABSOLUTE error 0.1-1.0 % of the median adapter's rms - below the median adapter's own absolute error (0.6-1.2 %) - -> bar 2 %
TOL_BF16 = dict(pred=3e-2, loss=2e-2, ta=5e-2, reg=5e-2, cos=0.9995, rel=3e-2, rows_cos=0.999, rows_rel=5e-2, disp_cos=0.95, rows_final=2e-2, ada_rel=0.12, ada_normal_rel=0.10, ada_small_abs=0.02)
TOL_FAITHFUL = dict(pred=3e-2, loss=8e-3, cos=0.9995, rel=3e-2, rows_cos=0.999, rows_rel=5e-2, ada_rel=0.12, ada_normal_rel=0.10, ada_small_abs=0.02)
TOL_FP32 = dict(pred=2e-3, loss=1e-3, ta=2e-3, reg=2e-3, cos=0.9999, rel=5e-3, rows_cos=0.9999, rows_rel=1e-2, disp_cos=0.99, rows_final=1e-3, ada_rel=2e-2)
TOL_FP32_FAITHFUL = dict(pred=3e-2, loss=2e-2, cos=0.99, rel=8e-2, rows_cos=0.985, rows_rel=0.2, ada_rel=0.45)     # (fp32 engine vs rounded oracle: the bf16 bars)
# the "trained-like" weight mode (_trained_like): no adapter is allowed to hide behind "its gradient is numerically nothing" - every adapter tensor's cosine is bounded
# (measured, round 6, profiles/r06_parity_report.json: all 1154 adapter tensors cos >= 0.9960 against the fp32 oracle / 0.9928 against the bf16-faithful one, the smallest gradient 1.8 % of
# the median adapter's rms; the peaked softmax amplifies what bf16 storage leaves on the flat vector - rel-L2 3.1 / 3.8 %, prediction 2.2 / 2.4 % of max-abs - so those bars sit wider here)
TRAINED_LIKE_EXTRA = dict(ada_min_cos=0.99, pred=4e-2, cos=0.998, rel=6e-2, ada_rel=0.2, ada_normal_rel=0.2)
TOL_TRAINED_LIKE = dict(TOL_BF16, **TRAINED_LIKE_EXTRA)
REPORT = {}        # case -> worst adapters etc., written to gpurun_out/parity_report.json when that directory exists


def _per_adapter(names, lora, got, flat_ref):
    """[(adjusted relative error, cosine, name.A|B|M, rms / median rms)] of every adapter tensor's gradient against its slice of the oracle's
    flat gradient, worst first.  adjusted relative error = |got - ref| / sqrt(|ref|^2 + (0.1 median)^2): the plain relative L2 error for
    every adapter with a normal-sized gradient; adapters whose gradient is (numerically) nothing - self-attention q / k at 64 tokens reach
    1 % of the median rms - are judged against a tenth of the median adapter instead of against their own noise."""
    res, off = [], 0
    for k in names:
        for t, tag, g in zip(lora[k], ("A", "B", "M"), got[k]):
            n = t.numel()
            ref = flat_ref[off:off + n]
            c, _ = _cos_rel(g.reshape(-1), ref)
            res.append([float((g.reshape(-1).float().cpu() - ref.float()).norm()) / math.sqrt(n), c, f"{k}.{tag}", float(ref.norm()) / math.sqrt(n)])
            off += n
    assert off == flat_ref.numel()
    med = sorted(x[3] for x in res)[len(res) // 2]
    # the two classes behind the adjusted figure, reported on their own: adapters with a normal-sized gradient (rms >= 0.1 median) are judged by their
    # plain relative L2 error; the rest - self-attention q / k adapters deep in the 1280-wide stacks, whose gradient is ~1 % of the median adapter's because the softmax of a
    # random-init model is near uniform - by their ABSOLUTE error in units of the median adapter's rms: what bf16 storage of dS leaves on them is noise of the size every
    # adapter carries, it only looks large against a gradient that is (numerically) nothing
    _per_adapter.classes = dict(worst_normal_rel=max((x[0] / x[3] for x in res if x[3] >= 0.1 * med), default=0.0),
                                worst_small_abs_over_median_rms=max((x[0] / med for x in res if x[3] < 0.1 * med), default=0.0),
                                n_small=sum(1 for x in res if x[3] < 0.1 * med), median_abs_over_median_rms=sorted(x[0] / med for x in res)[len(res) // 2])
    for x in res:
        x[0] = x[0] / math.sqrt(x[3] ** 2 + (0.1 * med) ** 2)
        x[3] = x[3] / med
    return sorted((tuple(x) for x in res), reverse=True)


def _check_first_step(tag, tol, names, lora, pred, got, rows, o, dora):
    """-> (report, failures): every first-step comparison against one oracle mode; the caller records the report, THEN raises."""
    fails = []
    err = float((pred - o["pred"]).abs().max()) / float(o["pred"].abs().max())
    if err > tol["pred"]:
        fails.append(f"[{tag}] prediction error {err}")
    flat = torch.cat([t.reshape(-1) for k in names for t in got[k]])
    cos, rel = _cos_rel(flat, o["lora_grads"])
    if not (cos >= tol["cos"] and rel <= tol["rel"]):
        fails.append(f"[{tag}] LoRA grads cos {cos} rel {rel}")
    per = _per_adapter(names, lora, got, o["lora_grads"])
    by_cos = sorted(per, key=lambda x: x[1])
    rep = dict(pred_err=err, lora_cos=cos, lora_rel=rel, worst_adapters=[dict(adj_rel=round(r, 4), cos=round(c, 5), name=n, rms_over_median=round(w, 4)) for r, c, n, w in per[:5]],
               median_adapter_rel=per[len(per) // 2][0], n_adapter_tensors=len(per), adapter_classes=dict(_per_adapter.classes),
               min_adapter_cos=by_cos[0][1], adapters_below_cos_0_99=sum(1 for x in per if x[1] < 0.99), smallest_rms_over_median=min(x[3] for x in per),
               lowest_cos_adapters=[dict(cos=round(c, 5), name=n, rms_over_median=round(w, 4), adj_rel=round(r, 4)) for r, c, n, w in by_cos[:5]])
    if "ada_min_cos" in tol and by_cos[0][1] < tol["ada_min_cos"]:
        fails.append(f"[{tag}] adapters below cos {tol['ada_min_cos']}: {by_cos[:5]}")
    if per[0][0] > tol["ada_rel"]:
        fails.append(f"[{tag}] worst adapter gradients (adjusted rel, cos, name, rms / median) {per[:5]}")
    cl = _per_adapter.classes
    if cl["worst_normal_rel"] > tol.get("ada_normal_rel", tol["ada_rel"]) or cl["worst_small_abs_over_median_rms"] > tol.get("ada_small_abs", 0.1 * tol["ada_rel"] * 1.5):
        fails.append(f"[{tag}] adapter classes {cl}")
    rr = []
    for got_r, ref_r in zip(rows, o["row_grads"]):
        cos, rel = _cos_rel(got_r, ref_r)
        rr.append((cos, rel))
        if not (cos >= tol["rows_cos"] and rel <= tol["rows_rel"]):
            fails.append(f"[{tag}] token-row grads cos {cos} rel {rel}")
    rep["token_rows"] = rr
    return rep, fails


def run_step_and_trajectory(version, B, h, sd, kinds, *, device, ops=None, act_dtype=torch.bfloat16, tol=TOL_BF16, rank=16, n_steps=6, dora=False,
                            tol_faithful=TOL_FAITHFUL, case=None):
    """(a) + (b) for one topology; shared with the CPU test of the same flow on the toy topologies (op emulation, fp32).
    dora: weight-decomposed adapters (use_dora: magnitudes trained too, no L1 penalty / weight decay, config.py:153-157)."""
    from oracle import step_ref as R
    from oracle import unet_ref as U
    from sd_lora_trainer_amd import topology
    cfg = U.CONFIGS[version]
    xl = cfg["addition"]
    w_ta = 2e-2            # token-attention weight raised from 3e-7 so that its gradient is visible in the comparison
    lr, lr_ti = 4e-4, 1e-3
    vc = Vocab(topology.CLIP_CONFIGS[kinds[0]]["vocab"])
    lora = U.init_lora(cfg, rank, seed=1, b_std=0.02)
    l1, wd = (0.0, 0.0) if dora else (0.03, 0.004)
    if dora:        # magnitudes = the weight norms at injection, perturbed so that the column factor differs from 1
        lora = U.init_dora_magnitudes(cfg, sd, lora, jitter=0.05, seed=5)
    hf = [_hf_clip(k, 11 + i) for i, k in enumerate(kinds)]
    batches = [_batch(cfg, B, h, 3, [10, 900, 500, 999], vc), _batch(cfg, B, h, 4, [700, 50, 300, 850], vc)]
    cuda = torch.device(device).type == "cuda"
    sync = torch.cuda.synchronize if cuda else (lambda: None)

    rt, unet, ts = _build_product(version, B, h, sd, lora, hf, rank, kinds=kinds, device=device, ops=ops, act_dtype=act_dtype, use_dora=dora, snr_gamma=5.0,
                                  l1_penalty=l1, weight_decay=wd, token_attention_loss_w=w_ta, ti_std_loss_w=0.01)
    ref = R.RefTrainer(cfg, sd, lora, text_models=hf, n_tokens=NTOK, train_ids=vc.train_ids, snr_gamma=5.0, l1_penalty=l1,
                       weight_decay=wd, token_attention_loss_w=w_ta, ti_std_loss_w=0.01)
    names = list(lora)
    n_enc = len(hf)
    traj = []
    for step in range(n_steps):
        b = batches[step % 2]
        tid = _set(ts, b, xl, h, n_enc)
        if step == 0 and tol_faithful is not None:      # the second yardstick, before the oracle's first optimizer step moves its parameters
            ob = ref.gradients(b["latent"], b["noise"], b["t"], b["mask"], lr_ti=lr_ti, ids=b["ids"], caption_token_lists=b["lists"], time_ids=tid, bf16_faithful=True)
        o = ref.step(b["latent"], b["noise"], b["t"], b["mask"], lr=lr, lr_ti=lr_ti, ids=b["ids"], caption_token_lists=b["lists"], time_ids=tid)
        if step == 0:
            # eager first step: everything the oracle exposes
            pred = ts.forward_backward().float().cpu().reshape(B, h, h, 4).permute(0, 3, 1, 2)
            sync()
            assert torch.isfinite(pred).all()
            got = unet.arena.export("grads")
            rows = [r.clone() for r in ts.ti.grad_rows]
            rep, fails = {}, []
            # the debug read-out (main.py:373-379): gradient norm of the UNet adapters and of every text encoder (= its trained token rows)
            gn = ts.grad_norms()
            assert abs(gn["unet"] - float(o["lora_grads"].norm())) <= tol["rel"] * float(o["lora_grads"].norm()), (gn, float(o["lora_grads"].norm()))
            for i, ref_r in enumerate(o["row_grads"]):
                assert abs(gn[f"text_encoder_{i}"] - float(ref_r.norm())) <= tol["rows_rel"] * float(ref_r.norm()), (gn, float(ref_r.norm()))
            rep["fp32_oracle"], f1 = _check_first_step("fp32 oracle", tol, names, lora, pred, got, rows, o, dora)
            fails += f1
            if tol_faithful is not None:
                rep["bf16_faithful_oracle"], f2 = _check_first_step("bf16-faithful oracle", tol_faithful, names, lora, pred, got, rows, ob, dora)
                fails += f2
                rep["bf16_faithful_oracle"]["loss_rel"] = abs(float(ts.loss) - ob["img_loss"]) / abs(ob["img_loss"])
                if rep["bf16_faithful_oracle"]["loss_rel"] > tol_faithful["loss"]:
                    fails.append(f"[bf16-faithful oracle] image loss {float(ts.loss)} vs {ob['img_loss']}")
            if case is not None:
                REPORT[case] = rep
                out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
                if os.path.isdir(out_dir):
                    import json
                    with open(os.path.join(out_dir, "parity_report.json"), "w") as fh:
                        json.dump(REPORT, fh, indent=1)
            assert not fails, "\n".join(fails)
            cos, rel = _cos_rel(torch.cat([t.reshape(-1) for k in names for t in got[k]]), o["lora_grads"])
            if dora:        # the magnitude gradients on their own (a small share of the flat vector above)
                gm_ref, off = [], 0
                for k in names:
                    nA, nB, nM = (t.numel() for t in lora[k])
                    gm_ref.append(o["lora_grads"][off + nA + nB: off + nA + nB + nM])
                    off += nA + nB + nM
                cos, rel = _cos_rel(torch.cat([got[k][2].reshape(-1) for k in names]), torch.cat(gm_ref))
                assert cos >= tol["cos"] and rel <= tol["rel"], f"DoRA magnitude grads cos {cos} rel {rel}"
            ts.sync_gradients()
            ts.set_hyper(lr, lr_ti)
            ts.optimizer_step()
        else:
            if step == 1 and cuda:
                ts.capture(warmup=1)            # from here on: hipGraph replays, as train() runs them
            ts.run(lr, lr_ti=lr_ti)
        sync()
        traj.append((float(ts.loss), o["img_loss"], float(ts.ta.loss), o["token_attention_loss"], float(ts.ti.reg_loss), o["reg"],
                     float(ts.l1_sum) / unet.arena.n, o.get("l1", 0.0)))
    for i, (l, lo, ta, tao, rg, rgo, l1, l1o) in enumerate(traj):
        assert abs(l - lo) <= tol["loss"] * abs(lo), f"step {i}: image loss {l} vs oracle {lo}; trajectory {traj}"
        assert abs(ta - tao) <= tol["ta"] * abs(tao), f"step {i}: token-attention loss {ta} vs {tao}; {traj}"
        assert abs(rg - rgo) <= tol["reg"] * abs(rgo) + 1e-7, f"step {i}: token regulariser {rg} vs {rgo}; {traj}"
        assert dora or abs(l1 - l1o) <= 1e-3 * abs(l1o), f"step {i}: L1 norm {l1} vs {l1o}"
    if n_steps < 4:           # first step only (in detail) + a graph replay or two: no trajectory to judge
        return traj
    # training moved the loss of the revisited batches (so the comparison above is not a comparison of constants)
    assert traj[n_steps - 2][1] < traj[0][1] and traj[n_steps - 1][1] < traj[1][1], traj
    # final state: the LoRA displacement and the token rows agree with the oracle's
    start = torch.cat([t.reshape(-1).float() for k in names for t in lora[k]])
    got = unet.arena.export("params")
    disp, disp_ref = torch.cat([t.reshape(-1) for k in names for t in got[k]]) - start, ref.lora_flat() - start
    cos, rel = _cos_rel(disp, disp_ref)
    # What the raw displacement cosine mixes: AdamW moves a coordinate by ~lr * m_hat / sqrt(v_hat) per step, which is +-lr for ANY gradient of
    # consistent sign however small, and sign noise for a coordinate whose gradient is numerically nothing.  So (1) the moments themselves - m is linear and v quadratic in the
    # gradients of the six steps, no division: they must agree like the gradients do - and (2) the displacement on the coordinates whose oracle update is decided,
    # |m_hat| / sqrt(v_hat) >= 0.5 (the last step moved them by at least lr / 2 in a definite direction), with the share of such coordinates in the report.
    m_ref, v_ref, k_adam = ref.adam_moments()
    gm, gv = unet.arena.export("m"), unet.arena.export("v")
    m_got, v_got = (torch.cat([t.reshape(-1) for k in names for t in d_[k]]) for d_ in (gm, gv))
    assert k_adam == n_steps
    m_cos, m_rel = _cos_rel(m_got, m_ref)
    v_cos, v_rel = _cos_rel(v_got, v_ref)
    ratio = (m_ref / (1 - 0.9 ** k_adam)).abs() / ((v_ref / (1 - 0.999 ** k_adam)).sqrt() + 1e-8)
    decided = ratio >= 0.5
    d_cos, d_rel = _cos_rel(disp[decided], disp_ref[decided])
    u_cos, _ = _cos_rel(disp[~decided], disp_ref[~decided])
    strong = ratio >= 0.9                       # (reported, not asserted: the coordinates whose six gradients all but agree in sign)
    s_cos, _ = _cos_rel(disp[strong], disp_ref[strong])
    moments = dict(m_cos=m_cos, m_rel=m_rel, v_cos=v_cos, v_rel=v_rel, decided_fraction=float(decided.float().mean()), decided_cos=d_cos, decided_rel=d_rel, undecided_cos=u_cos,
                   strongly_decided_fraction=float(strong.float().mean()), strongly_decided_cos=s_cos)
    if case is not None and case in REPORT:
        REPORT[case]["displacement_after_steps"] = dict(steps=n_steps, cos=cos, rel=rel, **moments)
        out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
        if os.path.isdir(out_dir):
            import json
            with open(os.path.join(out_dir, "parity_report.json"), "w") as fh:
                json.dump(REPORT, fh, indent=1)
    assert cos >= tol["disp_cos"], f"LoRA displacement after {n_steps} AdamW steps: cos {cos} rel {rel}"
    # measured (round 5, SDXL / SDXL + DoRA, profiles/r05_parity_report.json): m cos 0.99992 / rel 1.2 %, v cos 0.99993 / rel 1.2 % - the moments agree like the gradients do,
    # i.e. the optimizer state is the oracle's; decided coordinates 19.4 % of 25 M with cos 0.9929, the undecided rest 0.960 - which is where the raw figure (0.9707) comes from
    assert m_cos >= tol.get("m_cos", 0.9995) and v_cos >= tol.get("v_cos", 0.9995) and m_rel <= tol.get("m_rel", 3e-2) and v_rel <= tol.get("m_rel", 3e-2), f"AdamW moments after {n_steps} steps: {moments}"
    assert d_cos >= tol.get("decided_cos", 0.98) and moments["decided_fraction"] >= 0.1, f"LoRA displacement on the decided coordinates after {n_steps} steps: {moments}"
    for rows, table in zip(ts.ti.rows, ref.tables):
        cos, rel = _cos_rel(rows, table.detach()[-NTOK:])
        assert cos >= 0.999 and rel <= tol["rows_final"], f"token rows after {n_steps} steps: cos {cos} rel {rel}"
    return traj


def _case_step_and_trajectory(version, B, dora=False, h=32, n_steps=6, rank=16, case=None, trained_like=False, **kw):
    """(a) + (b): first step in detail, then 5 more optimizer steps; batches alternate between two injected ones so that the
    effect of training on a revisited batch is part of what is compared."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from oracle import unet_ref as U
    kinds = ["clip_l", "clip_g"] if U.CONFIGS[version]["addition"] else ["clip_l"]
    sd = _unet_state(version)
    run_step_and_trajectory(version, B, h, _trained_like(sd) if trained_like else sd, kinds, device="cuda:0", dora=dora, n_steps=n_steps, rank=rank, case=case, **kw)


def _case_full_size_properties(version, B, h):
    """(c): the BASELINE configs at their full size - no oracle (a CPU step would take minutes), size-independent properties."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from oracle import unet_ref as U
    cfg = U.CONFIGS[version]
    xl = cfg["addition"]
    rank = 16
    sd = _unet_state(version)
    lora = U.init_lora(cfg, rank, seed=1, b_std=0.02)
    hf = [_hf_clip("clip_l", 11), _hf_clip("clip_g", 12)] if xl else [_hf_clip("clip_l", 11)]
    rt, unet, ts = _build_product(version, B, h, sd, lora, hf, rank, token_attention_loss_w=3e-7)
    b = _batch(cfg, B, h, 3, [10, 900, 500, 999])
    _set(ts, b, xl, h, len(hf))
    pred = ts.forward_backward()
    torch.cuda.synchronize()
    assert torch.isfinite(pred).all() and torch.isfinite(unet.arena.grads).all() and float(unet.arena.grads.abs().max()) > 0
    assert all(torch.isfinite(r).all() and float(r.abs().max()) > 0 for r in ts.ti.grad_rows)
    g_eager, rows_eager, loss_eager = unet.arena.grads.clone(), [r.clone() for r in ts.ti.grad_rows], float(ts.loss)
    # bit-reproducible: a second pass from the same state leaves the same bits everywhere (round 5: the GroupNorm statistics of the 1280- / 2560-wide maps had two
    # writers per slot with different summation splits - the only nondeterminism of the step, tools/determinism_probe.py; every reduction runs in a fixed order)
    ts.forward_backward()
    torch.cuda.synchronize()
    assert torch.equal(unet.arena.grads, g_eager) and all(torch.equal(a, e) for a, e in zip(ts.ti.grad_rows, rows_eager)) and float(ts.loss) == loss_eager, \
        "two eager passes from the same state differ"
    ts.capture(warmup=1)
    p0, rows0 = unet.arena.params.clone(), ts.ti.params.clone()
    assert float(unet.arena.m.abs().max()) == 0.0 and ts.opt_step == 0           # capture is not training
    ts.run(1e-3, lr_ti=1e-3)
    torch.cuda.synchronize()
    assert abs(float(ts.loss) - loss_eager) <= 2e-3 * abs(loss_eager), (float(ts.loss), loss_eager)
    cos, rel = _cos_rel(unet.arena.grads, g_eager)
    assert torch.equal(unet.arena.grads, g_eager), f"graph replay vs eager LoRA gradients differ: cos {cos} rel {rel}"      # (round 1: 5e-2, rounds 2-4: 1e-2, round 5: the same bits)
    for a, e in zip(ts.ti.grad_rows, rows_eager):
        cos, rel = _cos_rel(a, e)
        assert torch.equal(a, e), f"graph replay vs eager token-row gradients differ: cos {cos} rel {rel}"
    # optimizer state advanced: moments non-zero, parameters moved by ~lr, L1 norm read-out equals mean|p| of the arena
    assert ts.opt_step == 1 and float(unet.arena.m.abs().max()) > 0 and float(unet.arena.v.abs().max()) > 0
    d = (unet.arena.params - p0).abs()
    assert 0 < float(d.max()) <= 1.2e-3 and float(d.mean()) > 1e-4
    assert float((ts.ti.params - rows0).abs().max()) > 0
    assert abs(float(ts.l1_sum) - float(p0.abs().sum())) <= 1e-3 * float(p0.abs().sum())
    losses = [ts.total_loss()]
    for i in range(7):
        ts.run(1e-3, lr_ti=1e-3)
        losses.append(ts.total_loss())
    assert all(math.isfinite(x) for x in losses) and losses[-1] < losses[0], losses
    # frozen token rows (ti lr == 0): bit-identical rows, LoRA still trains
    rows1 = ts.ti.params.clone()
    ts.run(1e-3, lr_ti=0.0)
    torch.cuda.synchronize()
    assert torch.equal(rows1, ts.ti.params)


# Full fine-tune, per parameter tensor: gradient cosine against oracle autograd >= FULLFT_TENSOR_COS for every tensor whose gradient is at least
# FULLFT_WEAK_RMS of the median tensor's rms.  The tensors below that - measured: only self-attention to_q / to_k weights deep in the 1280-wide
# stacks, whose gradient is 0.4-0.7 % of the median (the softmax of a random-init model is near uniform, so d loss / d scores nearly cancels) - sit on
# the bf16 noise of the dS that feeds them: floor FULLFT_WEAK_COS, they must match FULLFT_WEAK_NAMES, and every tensor below FULLFT_TENSOR_COS is
# listed BY NAME with its rms ratio in the parity report (profiles/r04_parity_report.json -> `*fullft*` -> below_099).  Measured: 32 x 32 batch 2 worst
# tensor 0.9979 (0 below 0.99); 64 x 64 batch 1: 15 such q / k tensors at 0.972-0.99 (rms ratio 0.004-0.007), everything else >= 0.99.
FULLFT_TENSOR_COS, FULLFT_WEAK_RMS, FULLFT_WEAK_COS = 0.99, 0.01, 0.95
FULLFT_WEAK_NAMES = (".attn1.to_q.weight", ".attn1.to_k.weight")


def _write_report():
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out_dir):
        import json
        with open(os.path.join(out_dir, "parity_report.json"), "w") as fh:
            json.dump(REPORT, fh, indent=1)


def _fullft_inputs(cfg, B, h, seed=3):
    g = torch.Generator().manual_seed(seed)
    latent = torch.randn(B, 4, h, h, generator=g) * cfg["scaling_factor"]
    noise = torch.randn(B, 4, h, h, generator=g)
    mask = (torch.rand(B, 1, h, h, generator=g) * 0.95 + 0.05).repeat(1, 4, 1, 1).contiguous()
    t = torch.tensor([10, 900, 500, 730][:B])
    ctx = torch.randn(B, 77, cfg["cross_dim"], generator=g)
    pooled = torch.randn(B, cfg["proj_class_in"] - 6 * cfg["addition_time_embed_dim"], generator=g)
    tid = torch.tensor([[1024., 1024, 0, 0, 8. * h, 8. * h]] * B)
    return latent, noise, mask, t, ctx, pooled, tid, {"text_embeds": pooled, "time_ids": tid}


def _fullft_compare(version, B, h, case):
    """One full fine-tune step of the real topology: the gradient of EVERY parameter tensor against autograd through the fp32 oracle - the flat
    2.57 G-vector (cos, rel L2) and every tensor on its own (cosine >= FULLFT_TENSOR_COS unless named in FULLFT_LOW_COS_OK; the ten worst go to
    the parity report).  Returns (TrainStep, WeightTrainer) for the caller's replay checks."""
    from oracle import unet_ref as U
    from sd_lora_trainer_amd import fullft, topology
    import sd_lora_trainer_amd.step as S
    import sd_lora_trainer_amd.unet as M
    from tests.test_fullft_cpu import oracle_grads
    cfg = U.CONFIGS[version]
    sd = _unet_state(version)
    latent, noise, mask, t, ctx, pooled, tid, add = _fullft_inputs(cfg, B, h)
    pred_o, loss_o, grads_o = oracle_grads(cfg, sd, latent, noise, t, mask, ctx, add)
    rt = M.Runtime("cuda:0", B)
    tr = fullft.WeightTrainer(rt)
    unet = M.UNet(rt, topology.CONFIGS[version], sd, trainer=tr)
    assert tr.n >= 2567463684
    ts = S.TrainStep(rt, unet, latent_hw=(h, h), snr_gamma=5.0)
    dv = lambda x: x.cuda() if x is not None else None  # noqa: E731
    ts.set_batch(dv(latent), dv(noise), dv(t), dv(mask), dv(ctx), dv(pooled), dv(tid))
    pred = ts.forward_backward().float().cpu().reshape(B, h, h, 4).permute(0, 3, 1, 2)
    torch.cuda.synchronize()
    perr = float((pred - pred_o).abs().max()) / float(pred_o.abs().max())
    got = tr.export("grads")
    num = den_a = den_b = dif = 0.0
    per = []
    for k in list(grads_o):            # streamed: the flat concatenation would be another 2 x 10 GB
        a, b_ = got[k].reshape(-1).double(), grads_o[k].reshape(-1).double()
        ab, aa, bb = float(a @ b_), float(a @ a), float(b_ @ b_)
        num, den_a, den_b, dif = num + ab, den_a + aa, den_b + bb, dif + float((a - b_) @ (a - b_))
        if b_.numel() >= 64 and bb > 0:
            per.append((ab / math.sqrt(aa * bb + 1e-300), k, b_.numel(), math.sqrt(bb / b_.numel())))
    cos, rel = num / math.sqrt(den_a * den_b), math.sqrt(dif / den_b)
    per.sort()
    med_rms = sorted(x[3] for x in per)[len(per) // 2]
    below = [dict(cos=round(c, 5), name=k, rms_over_median=round(r / med_rms, 4)) for c, k, n, r in per if c < FULLFT_TENSOR_COS]
    REPORT[case] = dict(below_099=below, B=B, latent=h, pred_err=perr, loss_rel=abs(float(ts.loss) - loss_o) / abs(loss_o), all_parameter_cos=cos, all_parameter_rel=rel, n_tensors=len(per),
                        n_tensors_below_0999=sum(1 for x in per if x[0] < 0.999), n_tensors_below_099=sum(1 for x in per if x[0] < 0.99),
                        worst_tensors=[dict(cos=round(c, 5), name=k, numel=n, rms_over_median=round(r / med_rms, 4)) for c, k, n, r in per[:10]])
    _write_report()
    assert perr <= 3e-2, perr
    assert abs(float(ts.loss) - loss_o) <= 1e-2 * abs(loss_o)
    assert cos >= 0.999 and rel <= 4e-2, f"all-parameter gradient: cos {cos} rel {rel}"
    low = [(c, k, r / med_rms) for c, k, _, r in per if c < FULLFT_TENSOR_COS and not (r / med_rms < FULLFT_WEAK_RMS and k.endswith(FULLFT_WEAK_NAMES) and c >= FULLFT_WEAK_COS)]
    assert not low, f"parameter tensors with gradient cosine < {FULLFT_TENSOR_COS} (cos, name, rms / median): {low[:10]}"
    del got, grads_o
    return ts, tr


def _case_fullft_real_sdxl_topology():
    """(d) cfg5's model: full fine-tune of the real SDXL UNet, batch 2 at a 32 x 32 latent - the gradient of every one of the
    2,567,463,684 parameters against autograd through the fp32 oracle; then graph replays train."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    ts, _ = _fullft_compare("sdxl", 2, 32, "sdxl-fullft-gradients-32x32-b2")
    ts.capture(warmup=1)
    losses = []
    for i in range(6):
        ts.run(2e-5)
        losses.append(float(ts.loss))
    assert all(x == x for x in losses) and losses[-1] < losses[0], losses


def _case_fullft_baseline_size():
    """cfg5 at its BASELINE size (SDXL 512 px = 64 x 64 latent, batch 4 per GPU, full_finetuning_example.json): one whole step - every parameter's
    gradient against oracle autograd at exactly that size - then the size-independent properties on the same job: the hipGraph replay reproduces
    the eager gradients, the optimizer state advances, the loss of the fixed batch goes down."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    ts, tr = _fullft_compare("sdxl", 4, 64, "sdxl-fullft-gradients-64x64-b4")
    nm = tr.n_mat                      # the matrix / conv weights (the vector tail - biases, norm affines - is cleared after the optimizer step)
    g_eager, loss_eager = tr.grads[:nm].clone(), float(ts.loss)
    assert torch.isfinite(tr.grads).all() and float(g_eager.abs().max()) > 0
    ts.capture(warmup=1)
    p0 = tr.params[:nm].clone()
    ts.run(2e-5)
    torch.cuda.synchronize()
    assert abs(float(ts.loss) - loss_eager) <= 2e-3 * abs(loss_eager), (float(ts.loss), loss_eager)
    ga = tr.grads[:nm]
    ab = aa = bb = 0.0
    for o in range(0, nm, 1 << 28):            # (2.57 G elements: torch.dot indexes with 32 bits)
        x, y = ga[o:o + (1 << 28)].double(), g_eager[o:o + (1 << 28)].double()
        ab, aa, bb = ab + float(x @ y), aa + float(x @ x), bb + float(y @ y)
    cosg = ab / math.sqrt(aa * bb)
    assert cosg >= 0.9999, f"graph replay vs eager weight gradients: cos {cosg}"
    del g_eager
    losses = [loss_eager]
    for i in range(6):
        ts.run(2e-5)
        losses.append(float(ts.loss))
    torch.cuda.synchronize()
    assert all(x == x for x in losses) and losses[-1] < losses[0], losses
    assert float((tr.params[:nm] - p0).abs().max()) > 0
    REPORT["sdxl-fullft-64x64-b4-properties"] = dict(losses=losses, graph_vs_eager_grad_cos=cosg)
    _write_report()


# Ordered so that each 10 GB weight state is built once: all SDXL cases, then all SD1.5 cases.
@pytest.mark.parametrize("case", ["sdxl-step-trajectory", "sdxl-trained-like-step", "sdxl-trained-like-full-size-step-parity", "sdxl-dora-step-trajectory", "sdxl-rank24-step", "sdxl-full-size", "sdxl-full-size-step-parity",
                                  "sdxl-fullft-gradients", "sdxl-fullft-baseline-size", "sd15-step-trajectory", "sd15-dora-step-trajectory", "sd15-rank64-step", "sd15-full-size",
                                  "sd15-full-size-step-parity"])
def test_real_topology(case):
    if case == "sdxl-step-trajectory":
        _case_step_and_trajectory("sdxl", 1, case=case)
    elif case == "sdxl-trained-like-step":         # peaked softmax, outlier channels, non-trivial norm gains: every adapter's gradient carries signal
        _case_step_and_trajectory("sdxl", 1, n_steps=2, case=case, trained_like=True, tol=TOL_TRAINED_LIKE, tol_faithful=dict(TOL_FAITHFUL, **TRAINED_LIKE_EXTRA))
    elif case == "sdxl-trained-like-full-size-step-parity":      # the same mode at cfg3's full size: a report run (~50 s of oracle time), not in the driver's suite
        if os.environ.get("SDLT_PARITY_EXTRA") != "1":
            pytest.skip("report run: SDLT_PARITY_EXTRA=1 (profiles/r06_parity_report.json)")
        _case_step_and_trajectory("sdxl", 1, h=128, n_steps=1, case=case, trained_like=True, tol=TOL_TRAINED_LIKE, tol_faithful=dict(TOL_FAITHFUL, **TRAINED_LIKE_EXTRA))
    elif case == "sdxl-full-size-step-parity":     # cfg3 at its FULL size (1024 px: 128 x 128 latent, batch 1): one whole step against the fp32 oracle
        _case_step_and_trajectory("sdxl", 1, h=128, n_steps=1, case=case)
    elif case == "sd15-full-size-step-parity":     # cfg2 at its FULL size (512 px: 64 x 64 latent, batch 4)
        _case_step_and_trajectory("sd15", 4, h=64, n_steps=1, case=case)
    elif case == "sdxl-dora-step-trajectory":      # use_dora on all 577 adapted layers
        _case_step_and_trajectory("sdxl", 1, dora=True, case=case)
    elif case == "sd15-dora-step-trajectory":      # the hyper-parameter sweep's variant (SD1.5 + use_dora, create_hyperparam_sweep.py:55,77)
        _case_step_and_trajectory("sd15", 4, dora=True, case=case)
    elif case == "sdxl-rank24-step":               # the sweep's ranks 24 / 64 (rank pads 32 / 64: K-grouped dX and batched K/V launches at the real widths)
        _case_step_and_trajectory("sdxl", 1, n_steps=2, rank=24, case=case)
    elif case == "sd15-rank64-step":
        _case_step_and_trajectory("sd15", 4, n_steps=2, rank=64, case=case)
    elif case == "sdxl-full-size":
        _case_full_size_properties("sdxl", 1, 128)
    elif case == "sdxl-fullft-gradients":
        _case_fullft_real_sdxl_topology()
    elif case == "sdxl-fullft-baseline-size":      # cfg5 at its BASELINE size (64 x 64 latent, batch 4): the whole step against oracle autograd + replay properties
        _case_fullft_baseline_size()
    elif case == "sd15-step-trajectory":
        _case_step_and_trajectory("sd15", 4, case=case)
    else:
        _case_full_size_properties("sd15", 4, 64)
