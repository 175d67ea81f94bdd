"""Pins the oracle (oracle/loss_ref.py) against golden vectors produced by the REFERENCE's own
functions (oracle/gen_golden.py).  CPU only."""
import os

import torch

from oracle import loss_ref as L


def _load(golden_dir, name):
    return torch.load(os.path.join(golden_dir, name), weights_only=False)


def test_alphas_cumprod_and_snr(golden_dir):
    g = _load(golden_dir, "diffusion_loss.pt")
    acp = L.ddpm_alphas_cumprod()
    assert torch.equal(acp, g["alphas_cumprod"])
    torch.testing.assert_close(L.compute_snr(acp, g["snr_t"]), g["snr"], rtol=1e-6, atol=0)


def test_diffusion_loss_matches_reference(golden_dir):
    g = _load(golden_dir, "diffusion_loss.pt")
    acp = g["alphas_cumprod"]
    assert len(g["cases"]) == 18
    for c in g["cases"]:
        got = L.diffusion_loss(c["pred"], c["noise"], c["noisy"], c["mask"], acp, c["t"],
                               snr_gamma=c["gamma"], prediction_type=c["ptype"])
        torch.testing.assert_close(got, c["loss"], rtol=2e-6, atol=1e-7)


def test_daam_stack_and_token_attention_loss(golden_dir):
    for c in _load(golden_dir, "token_attention.pt"):
        st = L.daam_stack(c["scores"], c["ratio"])
        torch.testing.assert_close(st, c["stacked"], rtol=1e-6, atol=1e-6)
        got = L.token_attention_loss(st, c["masks"], c["id_lists"], c["train_ids"])
        torch.testing.assert_close(got, c["loss"], rtol=1e-5, atol=1e-6)
        none = L.token_attention_loss(st, c["masks"], c["id_lists_none"], c["train_ids"])
        assert float(none) == float(c["loss_none"]) == 0.0


def test_ti_regularizers(golden_dir):
    for c in _load(golden_dir, "ti_regularizers.pt"):
        d = L.DistributionStats(c["table"])
        torch.testing.assert_close(d.std_loss(c["rows"]), c["std_loss"], rtol=1e-5, atol=1e-8)
        torch.testing.assert_close(d.cov_loss(c["rows"]), c["cov_loss"], rtol=1e-5, atol=1e-8)


def test_adamw_trajectory(golden_dir):
    g = _load(golden_dir, "adamw.pt")
    for tr in g["traj"]:
        p = tr["p0"].clone()
        m = torch.zeros_like(p)
        v = torch.zeros_like(p)
        for i, (gr, lr) in enumerate(zip(tr["grads"], tr["lrs"])):
            L.adamw_step(p, gr, m, v, i + 1, lr, weight_decay=tr["wd"])
            torch.testing.assert_close(p, tr["states"][i], rtol=1e-6, atol=1e-8)
    # rows-only update == full-table update with masked grads (wd = 0)   [SURVEY a17]
    n = g["n_tokens"]
    rows = g["table0"][-n:].clone()
    m = torch.zeros_like(rows)
    v = torch.zeros_like(rows)
    for i, gr in enumerate(g["table_grads"]):
        L.adamw_step(rows, gr[-n:], m, v, i + 1, 1e-3)
    torch.testing.assert_close(rows, g["table_final"][-n:], rtol=1e-6, atol=1e-8)
    assert torch.equal(g["table_final"][:-n], g["table0"][:-n])


def test_daam_processor_attention_math(golden_dir):
    """The oracle UNet's attention (+ head-summed scores) against the reference's processor."""
    import math
    for c in _load(golden_dir, "daam_processor.pt"):
        x = c["x"].clone().requires_grad_(True)
        ctx = c["ctx"].clone().requires_grad_(True)
        ws = [c[k].clone().requires_grad_(True) for k in ("wq", "wk", "wv", "wo")]
        B, N, C, H = c["B"], c["N"], c["C"], c["heads"]
        d = C // H
        q = (x @ ws[0].T).view(B, N, H, d).transpose(1, 2)
        k = (ctx @ ws[1].T).view(B, 77, H, d).transpose(1, 2)
        v = (ctx @ ws[2].T).view(B, 77, H, d).transpose(1, 2)
        s = q @ k.transpose(-1, -2) / math.sqrt(d)
        scores = s.sum(1)
        o = (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(B, N, C) @ ws[3].T + c["bo"]
        torch.testing.assert_close(o, c["out"], rtol=1e-4, atol=1e-5)
        torch.testing.assert_close(scores, c["scores"], rtol=1e-4, atol=1e-5)
        grads = torch.autograd.grad([o, scores], [x, ctx] + ws, [c["go"], c["gs"]])
        for got, key in zip(grads, ("gx", "gctx", "gwq", "gwk", "gwv", "gwo")):
            torch.testing.assert_close(got, c[key], rtol=1e-4, atol=1e-5)


def test_target_prompt_loss(golden_dir):
    """embedding_handler.py:288-318 (token warm-up objective): values and gradients from the reference's own method."""
    for c in _load(golden_dir, "target_prompt_loss.pt"):
        pe = c["prompt_embeds"].clone().requires_grad_(True)
        pp = c["pooled"].clone().requires_grad_(True) if c["pooled"] is not None else None
        loss = L.target_prompt_loss(pe, c["target"], pp, c["target_pooled"])
        torch.testing.assert_close(loss, c["loss"], rtol=1e-6, atol=1e-7)
        grads = torch.autograd.grad(loss, [pe] + ([pp] if pp is not None else []))
        torch.testing.assert_close(grads[0], c["d_prompt"], rtol=1e-5, atol=1e-9)
        if pp is not None:
            torch.testing.assert_close(grads[1], c["d_pooled"], rtol=1e-5, atol=1e-9)


def test_prompt_norm_regulariser(golden_dir):
    """loss.py:235-239 (cond_reg_w term): value, norm read-out and gradient from the reference's own method."""
    for c in _load(golden_dir, "prompt_norm.pt"):
        pe = c["prompt_embeds"].clone().requires_grad_(True)
        loss, val = L.prompt_norm_loss(pe, c["target"])
        torch.testing.assert_close(loss, c["loss"], rtol=1e-6, atol=1e-8)
        torch.testing.assert_close(val, c["value"], rtol=1e-6, atol=0)
        (gr,) = torch.autograd.grad(loss, pe)
        torch.testing.assert_close(gr, c["grad"], rtol=1e-5, atol=1e-10)
