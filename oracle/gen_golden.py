"""Generates tests/golden/*.pt by importing the REFERENCE's own functions (read-only tree at
/root/reference) in this container.  The reference's third-party imports that are not installed
here (diffusers, peft, prodigyopt, ujson) are replaced by attribute-permissive stub modules so the
reference's pure-torch functions can run on CPU.  Only input/output tensors are
written - no reference source or bytecode travels.

Run:  python oracle/gen_golden.py            (needs /root/reference; the GPU box never runs this)
"""
import os
import sys
import types

import torch

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def _install_stubs():
    sys.dont_write_bytecode = True

    class _Any(types.ModuleType):
        def __getattr__(self, k):
            if k.startswith("__"):
                raise AttributeError(k)
            return type(k, (), {})

    for name in ["diffusers", "diffusers.utils", "diffusers.utils.deprecation_utils", "diffusers.models",
                 "diffusers.models.attention_processor", "diffusers.models.lora", "diffusers.loaders",
                 "peft", "peft.utils", "prodigyopt", "ujson", "dotenv", "cv2", "openai", "mediapipe"]:
        if name not in sys.modules:
            sys.modules[name] = _Any(name)
    ap = sys.modules["diffusers.models.attention_processor"]
    ap.AttnProcessor2_0 = type("AttnProcessor2_0", (), {})
    ap.Attention = type("Attention", (), {})
    sys.modules["diffusers.utils"].deprecate = lambda *a, **k: None
    sys.modules["diffusers.utils.deprecation_utils"].deprecate = lambda *a, **k: None
    sys.path.insert(0, REF)


class _Sched:
    def __init__(self, alphas_cumprod, prediction_type):
        self.alphas_cumprod = alphas_cumprod
        self.config = types.SimpleNamespace(prediction_type=prediction_type, num_train_timesteps=1000)

    def get_velocity(self, sample, noise, timesteps):  # diffusers formula (3P), needed only for v-pred
        a = self.alphas_cumprod[timesteps] ** 0.5
        s = (1 - self.alphas_cumprod[timesteps]) ** 0.5
        shape = (-1,) + (1,) * (sample.dim() - 1)
        return a.view(shape) * noise - s.view(shape) * sample


def main():
    _install_stubs()
    os.makedirs(OUT, exist_ok=True)
    import trainer.loss as rloss
    import trainer.ti_cross_attn_loss as rdaam

    g = torch.Generator().manual_seed(1234)
    betas = torch.linspace(0.00085 ** 0.5, 0.012 ** 0.5, 1000, dtype=torch.float32) ** 2
    acp = torch.cumprod(1.0 - betas, dim=0)

    # ---- (i)+(ii) diffusion loss / SNR -------------------------------------------------------
    cases = []
    for B, hw in [(1, 16), (4, 8), (3, 12)]:
        for gamma in [None, 0.0, 5.0]:
            for ptype in ["epsilon", "v_prediction"]:
                pred = torch.randn(B, 4, hw, hw, generator=g)
                noise = torch.randn(B, 4, hw, hw, generator=g)
                noisy = torch.randn(B, 4, hw, hw, generator=g)
                mask = (torch.rand(B, 1, hw, hw, generator=g) * 0.95 + 0.05).repeat(1, 4, 1, 1)
                t = torch.randint(0, 1000, (B,), generator=g)
                if B == 4 and gamma == 5.0:
                    t = torch.tensor([10, 900, 0, 999])
                cfg = types.SimpleNamespace(snr_gamma=gamma)
                loss = rloss.compute_diffusion_loss(cfg, pred, noise, noisy, mask, _Sched(acp, ptype), t)
                cases.append(dict(pred=pred, noise=noise, noisy=noisy, mask=mask, t=t, gamma=gamma,
                                  ptype=ptype, loss=loss.detach().clone()))
    snr_t = torch.tensor([0, 1, 10, 100, 500, 900, 998, 999])
    snr = rloss.compute_snr(_Sched(acp, "epsilon"), snr_t)
    torch.save(dict(cases=cases, snr_t=snr_t, snr=snr, alphas_cumprod=acp), os.path.join(OUT, "diffusion_loss.pt"))

    # ---- (iii) DAAM processor fwd + grads ------------------------------------------------------
    class _Attn:
        spatial_norm = None
        group_norm = None
        norm_cross = False
        residual_connection = False
        rescale_output_factor = 1.0

        def __init__(self, C, D, heads):
            self.heads = heads
            self.to_q = torch.nn.Linear(C, C, bias=False)
            self.to_k = torch.nn.Linear(D, C, bias=False)
            self.to_v = torch.nn.Linear(D, C, bias=False)
            self.to_out = [torch.nn.Linear(C, C), torch.nn.Identity()]

    daam_cases = []
    for (B, N, C, D, H) in [(2, 64, 64, 48, 2), (1, 144, 64, 32, 1)]:
        torch.manual_seed(7 + N)
        attn = _Attn(C, D, H)
        proc = rdaam.DAAMLossAttnProcessor2_0(name="x")
        x = torch.randn(B, N, C, generator=g, requires_grad=True)
        ctx = torch.randn(B, 77, D, generator=g, requires_grad=True)
        out = proc(attn, x, encoder_hidden_states=ctx)
        scores = proc.cross_attention_scores
        go = torch.randn(out.shape, generator=g)
        gs = torch.randn(scores.shape, generator=g) * 0.1
        params = [attn.to_q.weight, attn.to_k.weight, attn.to_v.weight, attn.to_out[0].weight]
        grads = torch.autograd.grad([out, scores], [x, ctx] + params, [go, gs])
        daam_cases.append(dict(
            B=B, N=N, C=C, D=D, heads=H, x=x.detach(), ctx=ctx.detach(),
            wq=attn.to_q.weight.detach(), wk=attn.to_k.weight.detach(), wv=attn.to_v.weight.detach(),
            wo=attn.to_out[0].weight.detach(), bo=attn.to_out[0].bias.detach(),
            out=out.detach(), scores=scores.detach(), go=go, gs=gs,
            gx=grads[0], gctx=grads[1], gwq=grads[2], gwk=grads[3], gwv=grads[4], gwo=grads[5]))
    torch.save(daam_cases, os.path.join(OUT, "daam_processor.pt"))

    # ---- (iv)+(v) DAAM stack + token-attention loss -------------------------------------------
    class _P:
        pass

    tok_cases = []
    for (B, sizes, ratio) in [(2, [(16, 16), (8, 8), (16, 16)], 1.0), (1, [(8, 16), (4, 8)], 2.0)]:
        procs = []
        for (h, w) in sizes:
            p = _P()
            p.name = f"l{len(procs)}"
            p.cross_attention_scores = torch.randn(B, h * w, 77, generator=g) * 3.0
            procs.append(p)
        dl = rdaam.DAAMLoss(procs)
        stacked = dl.process_and_stack_attention_scores(ratio)
        H0, W0 = sizes[0]
        masks = (torch.rand(B, 1, H0 * 2, W0 * 2, generator=g) * 0.95 + 0.05).repeat(1, 4, 1, 1)
        train_ids = [49408, 49409, 49410]
        id_lists = [[49406, 320, 1125, 539] + train_ids + [49407],
                    [49406, 320, 2368, 49407]][:B]          # second caption lacks the TI tokens
        captions = [f"c{i}" for i in range(B)]

        class _Tok:
            def encode(self, c):
                return id_lists[int(c[1:])]

        pipe = types.SimpleNamespace(tokenizer=_Tok())
        eh = types.SimpleNamespace(train_ids=train_ids)
        loss = rloss.compute_token_attention_loss(pipe, eh, captions, masks, dl)
        # also the "no caption contains the TI tokens" branch
        id_lists_none = [[49406, 320, 49407]] * B
        saved = id_lists
        id_lists = id_lists_none
        loss_none = rloss.compute_token_attention_loss(pipe, eh, captions, masks, dl)
        id_lists = saved
        tok_cases.append(dict(scores=[p.cross_attention_scores for p in procs], ratio=ratio, stacked=stacked,
                              masks=masks, id_lists=id_lists, train_ids=train_ids, loss=loss,
                              id_lists_none=id_lists_none, loss_none=loss_none))
    torch.save(tok_cases, os.path.join(OUT, "token_attention.pt"))

    # ---- (vi) std / covariance regularisers ----------------------------------------------------
    reg_cases = []
    for (V, Dm) in [(512, 48), (1000, 64)]:
        table = torch.randn(V, Dm, generator=g) * (0.01 + 0.01 * torch.rand(V, 1, generator=g))
        rows = torch.randn(3, Dm, generator=g) * 0.02
        dl = rloss.DistributionLoss(table)
        reg_cases.append(dict(table=table, rows=rows, std_loss=dl.compute_std_loss(rows),
                              cov_loss=dl.compute_covariance_loss(rows)))
    torch.save(reg_cases, os.path.join(OUT, "ti_regularizers.pt"))

    # ---- (vii) AdamW trajectories (torch.optim.AdamW, optimizer.py:18) -------------------------
    traj = []
    for (wd, n) in [(0.004, 257), (0.0, 64)]:
        p0 = torch.randn(n, generator=g) * 0.1
        grads = [torch.randn(n, generator=g) * 0.01 for _ in range(5)]
        lrs = [5e-5, 7e-5, 1e-4, 1.3e-4, 2e-4]
        p = torch.nn.Parameter(p0.clone())
        opt = torch.optim.AdamW([p], lr=1e-4, weight_decay=wd)
        states = []
        for gr, lr in zip(grads, lrs):
            opt.param_groups[0]["lr"] = lr
            p.grad = gr.clone()
            opt.step()
            states.append(p.detach().clone())
        traj.append(dict(p0=p0, grads=grads, lrs=lrs, wd=wd, states=states))
    # full-table-with-masked-grads == rows-only equivalence (optimizer.py:113-150 + main.py:368-371)
    table = torch.nn.Parameter(torch.randn(40, 16, generator=g))
    t0 = table.detach().clone()
    opt = torch.optim.AdamW([table], lr=1e-3, weight_decay=0.0)
    tgrads = [torch.randn(40, 16, generator=g) for _ in range(4)]
    for gr in tgrads:
        table.grad = gr.clone()
        table.grad.data[:-3, :] *= 0.0
        opt.step()
    torch.save(dict(traj=traj, table0=t0, table_grads=tgrads, table_final=table.detach().clone(), n_tokens=3),
               os.path.join(OUT, "adamw.pt"))

    # ---- (ix) TrainingConfig derived-field snapshots for every shipped train_configs/*.json -----------
    import glob
    import json
    import tempfile
    import trainer.config as rconfig
    snaps = {}
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)                      # the reference mkdirs output_dir relative to the cwd
        try:
            for f in sorted(glob.glob(os.path.join(REF, "train_configs", "*.json"))):
                c = rconfig.TrainingConfig.from_json(f)
                d = c.dict()
                for k in ("output_dir", "start_time", "seed", "device", "job_time"):     # time / host dependent
                    d.pop(k, None)
                snaps[os.path.basename(f)] = {"input": json.load(open(f)), "derived": d}
        finally:
            os.chdir(cwd)
    with open(os.path.join(OUT, "config_snapshots.json"), "w") as fh:
        json.dump(snaps, fh, indent=1, sort_keys=True)
    gen_target_prompt_loss()
    gen_dataset()
    gen_blend_conditions()
    gen_prompt_norm()


def gen_prompt_norm():
    """(xiv) ConditioningRegularizer._compute_regularization_loss (trainer/loss.py:235-239) with autograd gradients."""
    _install_stubs()
    import types as _t
    import trainer.loss as rloss
    g = torch.Generator().manual_seed(17)
    cases = []
    for B, D, target in [(1, 64, 34.5), (3, 48, 27.8)]:
        pe = (torch.randn(B, 77, D, generator=g) * 3).requires_grad_(True)
        me = _t.SimpleNamespace(target_norm=target)
        loss, val = rloss.ConditioningRegularizer._compute_regularization_loss(me, pe)
        (gr,) = torch.autograd.grad(loss, pe)
        cases.append(dict(prompt_embeds=pe.detach(), target=target, loss=loss.detach(), value=val.detach(), grad=gr))
    torch.save(cases, os.path.join(OUT, "prompt_norm.pt"))


def gen_blend_conditions():
    """(xiii) blend_conditions (trainer/inference.py:180-228): SDXL 4-tuples and SD1.5 2-tuples, default and explicit token scale."""
    _install_stubs()
    import contextlib
    import io
    import trainer.inference as rinf
    g = torch.Generator().manual_seed(41)
    cases = []
    for sdxl in (True, False):
        for lora_scale, token_scale in [(0.75, None), (0.85, None), (0.3, 0.0), (1.0, None)]:
            mk = lambda *s: torch.randn(*s, generator=g)  # noqa: E731
            e1 = (mk(1, 77, 32), mk(1, 77, 32)) + ((mk(1, 16), mk(1, 16)) if sdxl else ())
            e2 = (mk(1, 77, 32), mk(1, 77, 32)) + ((mk(1, 16), mk(1, 16)) if sdxl else ())
            with contextlib.redirect_stdout(io.StringIO()):
                out, ts = rinf.blend_conditions(e1, e2, lora_scale, token_scale=token_scale)
            cases.append(dict(e1=e1, e2=e2, lora_scale=lora_scale, token_scale_in=token_scale, out=out, token_scale=float(ts)))
    torch.save(cases, os.path.join(OUT, "blend_conditions.pt"))


def gen_dataset():
    """(xii) PreprocessedDataset (trainer/dataset.py:31-193) run on a small on-disk dataset with a duck-typed VAE encoder
    (returns a given posterior) and image processor: processed captions, latent-resolution masks and the fetched latents
    for a seeded RNG.  Inputs (mask images as uint8 arrays, captions, posterior moments) travel with the outputs."""
    _install_stubs()
    import tempfile
    import types as _t
    import numpy as np
    import pandas as pd
    from PIL import Image
    import trainer.dataset as rds
    rng = np.random.RandomState(5)
    g = torch.Generator().manual_seed(9)
    n, size, lat = 3, [64, 48], (6, 8)                 # size = [w, h]; latent = [h/8, w/8]
    posts = [torch.randn(1, 8, *lat, generator=g) for _ in range(n)]
    posts[1][:, 4:] = 25.0                             # exercises the logvar clamp at 20
    masks_u8 = [rng.randint(0, 256, (40, 52)).astype(np.uint8) for _ in range(n)]
    captions = ["A photo of TOK, on a Beach", float("nan"), "tok and Tok with a dog"]
    sub = {"TOK": "<s0><s1><s2>"}

    class Dist:                                        # diffusers DiagonalGaussianDistribution (3P), restated for the fake encoder
        def __init__(self, p):
            self.mean, lv = torch.chunk(p, 2, dim=1)
            self.std = torch.exp(0.5 * torch.clamp(lv, -30.0, 20.0))

        def sample(self):
            return self.mean + self.std * torch.randn(self.mean.shape)

    class Enc:
        dtype, device = torch.float32, torch.device("cpu")
        config = _t.SimpleNamespace(scaling_factor=0.13025)

        def __init__(self):
            self.i = 0

        def encode(self, image):
            d = Dist(posts[self.i])
            self.i += 1
            return _t.SimpleNamespace(latent_dist=d)

    pipe = _t.SimpleNamespace(image_processor=_t.SimpleNamespace(
        preprocess=lambda pil: torch.from_numpy(np.array(pil).astype(np.float32) / 127.5 - 1.0).permute(2, 0, 1).unsqueeze(0)))
    with tempfile.TemporaryDirectory() as tmp:
        rows = []
        for i in range(n):
            Image.fromarray(rng.randint(0, 256, (40, 52, 3)).astype(np.uint8)).save(os.path.join(tmp, f"{i}.png"))
            Image.fromarray(masks_u8[i]).save(os.path.join(tmp, f"{i}_mask.png"))
            rows.append(dict(image_path=f"{i}.png", mask_path=f"{i}_mask.png", caption=captions[i]))
        pd.DataFrame(rows).to_csv(os.path.join(tmp, "captions.csv"), index=False)
        torch.manual_seed(123)
        ds = rds.PreprocessedDataset(tmp, pipe, Enc(), size=size, substitute_caption_map=sub)
        torch.manual_seed(321)
        items = [ds[i] for i in range(n)]
    torch.save(dict(posteriors=posts, masks_u8=masks_u8, captions_in=["" if c != c else c for c in captions], nan_index=1,
                    substitute=sub, size=size, scaling_factor=0.13025, fetch_seed=321,
                    captions=[it[0] for it in items], latents=[it[1] for it in items], masks=[it[2] for it in items]),
               os.path.join(OUT, "dataset.pt"))


def gen_target_prompt_loss():
    """(xi) TokenEmbeddingsHandler.compute_target_prompt_loss (embedding_handler.py:288-318), the objective of the
    token warm-up (pre_optimize_token_embeddings, :321-399): called unbound on a stand-in `self` that already holds the
    encoded target, so no pipeline is needed.  Values and autograd gradients w.r.t. the prompt embeddings."""
    _install_stubs()
    import types as _t
    import trainer.embedding_handler as reh
    g = torch.Generator().manual_seed(77)
    cases = []
    for B, D, P in [(1, 32, None), (2, 48, 24), (3, 256, 128)]:
        pe = torch.randn(B, 77, D, generator=g).requires_grad_(True)
        tpe = torch.randn(1, 77, D, generator=g)
        ppe = torch.randn(B, P, generator=g).requires_grad_(True) if P else None
        tppe = torch.randn(1, P, generator=g) if P else None
        me = _t.SimpleNamespace(target_prompt="x", target_prompt_embeds=tpe, target_pooled_prompt_embeds=tppe)
        loss = reh.TokenEmbeddingsHandler.compute_target_prompt_loss(me, "x", pe, ppe, None, None)
        grads = torch.autograd.grad(loss, [pe] + ([ppe] if P else []))
        cases.append(dict(prompt_embeds=pe.detach(), target=tpe, pooled=ppe.detach() if P else None, target_pooled=tppe,
                          loss=loss.detach(), d_prompt=grads[0], d_pooled=grads[1] if P else None))
    torch.save(cases, os.path.join(OUT, "target_prompt_loss.pt"))

    print("golden fixtures written to", os.path.normpath(OUT))
    for f in sorted(os.listdir(OUT)):
        print(" ", f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
