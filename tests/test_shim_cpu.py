"""the reference-shaped loop body (main.py:311-382) runs UNCHANGED on the call-compatible shim
(sd_lora_trainer_amd.shim.UNetModule): `unet(noisy, t, encoder_hidden_states=..., timestep_cond=None, added_cond_kwargs=...,
return_dict=False)[0]`, a torch-side loss, `loss.backward()`, `torch.optim.AdamW(unet.parameters())`, `optimizer.step()`.
Here on CPU through the op emulation against the fp32 oracle; tests/test_shim_gpu.py drives the same body over the HIP kernels."""
import pytest
import torch

from oracle import loss_ref as L
from oracle import unet_ref as U
from sd_lora_trainer_amd import shim
from tests import emu_ops
import sd_lora_trainer_amd.unet as unet_mod


def reference_loop_body(unet, optimizer, noise_acp, vae_latent, mask, prompt_embeds, pooled, add_time_ids, noise, timesteps, snr_gamma=5.0):
    """main.py:326-382 with the tensors the loop has at that point (noise / timesteps injected instead of drawn)."""
    noisy_latent = L.add_noise(noise_acp, vae_latent, noise, timesteps)
    model_pred = unet(noisy_latent, timesteps, encoder_hidden_states=prompt_embeds, timestep_cond=None,
                      added_cond_kwargs={"text_embeds": pooled, "time_ids": add_time_ids}, return_dict=False)[0]
    loss = L.diffusion_loss(model_pred, noise, noisy_latent, mask, noise_acp, timesteps, snr_gamma=snr_gamma)
    loss.backward()
    grads = [p.grad.clone() for p in unet.parameters()]
    optimizer.step()
    optimizer.zero_grad()
    return model_pred.detach(), float(loss.detach()), grads


def run_shim_vs_oracle(version, B, h, rt, tol, steps=4, sd=None, dora=False):
    cfg = U.CONFIGS[version]
    if sd is None:
        sd = {k: v.to(torch.bfloat16).float() for k, v in U.init_unet_state(cfg, seed=0).items()}
    lora = U.init_lora(cfg, 8, seed=1, b_std=0.03)
    if dora:          # LoraConfig(use_dora=True): (A, B, magnitude) per module, magnitudes off their initial value
        lora = U.init_dora_magnitudes(cfg, sd, lora, jitter=0.05, seed=7)
    npt = 3 if dora else 2
    dev = rt.device
    g = torch.Generator().manual_seed(5)
    latent = (torch.randn(B, 4, h, h, generator=g) * cfg["scaling_factor"]).to(dev)
    mask = (torch.rand(B, 1, h, h, generator=g) * 0.95 + 0.05).repeat(1, 4, 1, 1).to(dev)
    pe = torch.randn(B, 77, cfg["cross_dim"], generator=g).to(dev).requires_grad_(True)
    pooled = tid = None
    if cfg["addition"]:
        pooled = torch.randn(B, cfg["proj_class_in"] - 6 * cfg["addition_time_embed_dim"], generator=g).to(dev).requires_grad_(True)
        tid = torch.tensor([[1024., 1024, 0, 0, 8. * h, 8. * h]] * B).to(dev)
    acp = L.ddpm_alphas_cumprod().to(dev)

    unet = shim.get_peft_model(version, sd, shim.LoraConfig(r=8, lora_alpha=8.0, use_dora=dora), batch_size=B, runtime=rt)
    assert unet.device == rt.device and len(list(unet.parameters())) == npt * len(lora)
    unet.unet.arena.load(lora)
    unet.requires_grad_(True)
    wd = 0.0 if dora else 0.004
    opt = torch.optim.AdamW(list(unet.parameters()), lr=1e-3, weight_decay=wd)
    # oracle twin: same loop body over the oracle's functional UNet
    o_params = {k: tuple(t.clone().requires_grad_(True) for t in v) for k, v in lora.items()}
    o_opt = torch.optim.AdamW([t for ab in o_params.values() for t in ab], lr=1e-3, weight_decay=wd)
    o_pe = pe.detach().cpu().clone().requires_grad_(True)
    o_pooled = pooled.detach().cpu().clone().requires_grad_(True) if pooled is not None else None

    class OracleUNet:
        def parameters(self):
            return iter([t for ab in o_params.values() for t in ab])

        def __call__(self, sample, timestep, encoder_hidden_states=None, timestep_cond=None, added_cond_kwargs=None, return_dict=False):
            add = added_cond_kwargs if cfg["addition"] else None
            return (U.unet_forward(cfg, sd, sample, timestep, encoder_hidden_states, add, lora=o_params),)
    losses = []
    for step in range(steps):
        gs = torch.Generator().manual_seed(100 + step)
        noise = torch.randn(B, 4, h, h, generator=gs)
        t = torch.randint(0, 1000, (B,), generator=gs)
        pred, loss, grads = reference_loop_body(unet, opt, acp, latent, mask, pe, pooled, tid, noise.to(dev), t.to(dev))
        pred_o, loss_o, grads_o = reference_loop_body(OracleUNet(), o_opt, acp.cpu(), latent.cpu(), mask.cpu(), o_pe, o_pooled, tid.cpu() if tid is not None else None, noise, t)
        assert float((pred.cpu() - pred_o).abs().max()) <= tol["pred"] * float(pred_o.abs().max()), f"step {step}: prediction"
        assert abs(loss - loss_o) <= tol["loss"] * abs(loss_o), (step, loss, loss_o)
        # conv adapters: the module's parameters are the engine's 2-D (tap-major) layout; compare through the peft-layout export
        ga, gb = torch.cat([x.reshape(-1).cpu() for x in grads]), torch.cat([x.reshape(-1) for x in grads_o])
        assert ga.numel() == gb.numel()
        if step == 0:
            exp = unet.unet.arena.export("grads")
            flat = torch.cat([t_.reshape(-1) for k in lora for t_ in exp[k]])
            cos = float(flat.double() @ gb.double() / (flat.double().norm() * gb.double().norm()))
            assert cos >= tol["cos"], f"LoRA gradient cosine {cos}"
            ge, geo = pe.grad.cpu(), o_pe.grad
            cos = float(ge.reshape(-1).double() @ geo.reshape(-1).double() / (ge.double().norm() * geo.double().norm()))
            assert cos >= tol["cos"], f"encoder_hidden_states gradient cosine {cos}"
            if pooled is not None:
                gp, gpo = pooled.grad.cpu(), o_pooled.grad
                cos = float(gp.reshape(-1).double() @ gpo.reshape(-1).double() / (gp.double().norm() * gpo.double().norm()))
                assert cos >= tol["cos"], f"text_embeds gradient cosine {cos}"
        pe.grad = None
        o_pe.grad = None
        if pooled is not None:
            pooled.grad = None
            o_pooled.grad = None
        losses.append((loss, loss_o))
    # the torch optimizer moved the engine's adapters: exported weights equal the oracle twin's
    got = unet.get_peft_model_state_dict()
    for k, (A, Bm, *mg) in o_params.items():
        a, b_ = got[f"base_model.model.{k}.lora_A.weight"], got[f"base_model.model.{k}.lora_B.weight"]
        assert a.shape == A.shape and b_.shape == Bm.shape
        assert float((a - A.detach()).abs().max()) <= tol["param"] and float((b_ - Bm.detach()).abs().max()) <= tol["param"], k
        if mg:
            m_ = got[f"base_model.model.{k}.lora_magnitude_vector"]
            assert m_.shape == mg[0].shape and float((m_ - mg[0].detach()).abs().max()) <= tol["param"], k
    return unet, losses


@pytest.mark.parametrize("version,B,dora", [("tinyxl", 1, False), ("tiny15", 2, False), ("tinyxl", 1, True)])
def test_reference_loop_body_on_shim_cpu(version, B, dora, tmp_path):
    rt = unet_mod.Runtime("cpu", B, act_dtype=torch.float32, ops=emu_ops)
    unet, losses = run_shim_vs_oracle(version, B, 16, rt, dict(pred=2e-3, loss=1e-3, cos=0.9999, param=3e-4), dora=dora)
    unet.save_pretrained(str(tmp_path / "ad"))
    import json
    import os
    from safetensors.torch import load_file
    assert json.load(open(tmp_path / "ad" / "adapter_config.json"))["r"] == 8
    sd = load_file(os.path.join(tmp_path / "ad", "adapter_model.safetensors"))
    assert any(k.endswith("conv2.lora_A.weight") and v.dim() == 4 for k, v in sd.items()) and all(k.startswith("base_model.model.") for k in sd)
    # processor seam: per hooked attn2 layer a `.cross_attention_scores` [B, N, 77] after a forward with keep_daam_maps
    unet.keep_daam_maps = True
    cfg = U.CONFIGS[version]
    add = {"text_embeds": torch.zeros(B, cfg["proj_class_in"] - 6 * cfg["addition_time_embed_dim"]), "time_ids": torch.tensor([[1024., 1024, 0, 0, 128, 128]] * B)} if cfg["addition"] else None
    with torch.no_grad():
        ehs = torch.randn(B, 77, cfg["cross_dim"])
        x = torch.randn(B, 4, 16, 16)
        t = torch.tensor([500] * B)
        unet(x, t, encoder_hidden_states=ehs, added_cond_kwargs=add)
        _, daam = U.unet_forward(cfg, {k: v.to(torch.bfloat16).float() for k, v in U.init_unet_state(cfg, seed=0).items()}, x, t, ehs, add,
                                 lora=dict(unet.unet.arena.export()), return_daam=True)
    assert len(unet.daam_processors) == len(daam) > 0
    for proc, (name, s) in zip(unet.daam_processors, daam):
        assert proc.name == name + ".processor"
        torch.testing.assert_close(proc.cross_attention_scores, s, rtol=2e-3, atol=2e-3)


def run_shim_token_attention(version, B, h, rt, tol, sd=None):
    """SURVEY 8b seam 2: the score maps of the hooked attn2 layers are autograd outputs of the module call, so the reference-shaped
    token-attention loss (DAAMLoss stack + compute_token_attention_loss, ti_cross_attn_loss.py:239-268 / loss.py:10-80, here in their
    pinned restatement oracle/loss_ref.py) runs on `processor.cross_attention_scores` WITH gradient: image loss + w * token-attention loss,
    `loss.backward()`; LoRA gradients and the gradient w.r.t. the text conditioning against autograd through the oracle UNet."""
    cfg = U.CONFIGS[version]
    if sd is None:
        sd = {k: v.to(torch.bfloat16).float() for k, v in U.init_unet_state(cfg, seed=0).items()}
    lora = U.init_lora(cfg, 8, seed=1, b_std=0.03)
    dev = rt.device
    g = torch.Generator().manual_seed(9)
    latent = torch.randn(B, 4, h, h, generator=g) * cfg["scaling_factor"]
    noise = torch.randn(B, 4, h, h, generator=g)
    mask = (torch.rand(B, 1, h, h, generator=g) > 0.5).float().repeat(1, 4, 1, 1) * 0.9 + 0.05
    t = torch.randint(0, 1000, (B,), generator=g)
    pe = torch.randn(B, 77, cfg["cross_dim"], generator=g)
    pooled = torch.randn(B, cfg["proj_class_in"] - 6 * cfg["addition_time_embed_dim"], generator=g) if cfg["addition"] else None
    tid = torch.tensor([[1024., 1024, 0, 0, 8. * h, 8. * h]] * B) if cfg["addition"] else None
    train_ids = [900, 901, 902]
    lists = [[1, 5, 6] + (train_ids if b % 2 == 0 else [7]) + [8, 9, 2] for b in range(B)]       # one caption without the trigger tokens
    acp = L.ddpm_alphas_cumprod()
    w_ta = 5e-2            # (3e-7 in the reference's config: raised so that the side output's gradient is a visible share of the total)
    noisy = L.add_noise(acp, latent, noise, t)

    def total_loss(pred, maps, msk):
        loss = L.diffusion_loss(pred, noise.to(pred.device), noisy.to(pred.device), msk, acp.to(pred.device), t.to(pred.device), snr_gamma=5.0)
        return loss + w_ta * L.token_attention_loss(L.daam_stack(maps, 1.0), msk, lists, train_ids)

    # oracle
    o_params = {k: tuple(x.clone().requires_grad_(True) for x in v) for k, v in lora.items()}
    o_pe = pe.clone().requires_grad_(True)
    add = {"text_embeds": pooled, "time_ids": tid} if cfg["addition"] else None
    pred_o, daam_o = U.unet_forward(cfg, sd, noisy, t, o_pe, add, lora=o_params, return_daam=True)
    loss_o = total_loss(pred_o, [s for _, s in daam_o], mask)
    flat_params = [x for ab in o_params.values() for x in ab]
    grads_o = torch.autograd.grad(loss_o, flat_params + [o_pe])
    # plain image loss only, to show that the side output's gradient matters in this comparison
    g_img = torch.autograd.grad(L.diffusion_loss(U.unet_forward(cfg, sd, noisy, t, o_pe, add, lora=o_params), noise, noisy, mask, acp, t, snr_gamma=5.0), flat_params)

    # product: the call-compatible module, reference-shaped code on its processors' maps
    unet = shim.get_peft_model(version, sd, shim.LoraConfig(r=8, lora_alpha=8.0), batch_size=B, runtime=rt)
    unet.unet.arena.load(lora)
    unet.requires_grad_(True)
    unet.keep_daam_maps = True
    pe_d = pe.to(dev).requires_grad_(True)
    pred = unet(noisy.to(dev), t.to(dev), encoder_hidden_states=pe_d, timestep_cond=None,
                added_cond_kwargs={"text_embeds": pooled.to(dev), "time_ids": tid.to(dev)} if cfg["addition"] else None, return_dict=False)[0]
    maps = [proc.cross_attention_scores for proc in unet.daam_processors]
    assert len(maps) == len(daam_o) > 0 and all(m.requires_grad for m in maps)
    loss = total_loss(pred, maps, mask.to(dev))
    loss.backward()
    assert abs(float(loss) - float(loss_o)) <= tol["loss"] * abs(float(loss_o)), (float(loss), float(loss_o))
    exp = unet.unet.arena.export("grads")
    flat = torch.cat([x.reshape(-1) for k in lora for x in exp[k]]).double().cpu()
    ref = torch.cat([x.reshape(-1) for x in grads_o[:-1]]).double()
    img = torch.cat([x.reshape(-1) for x in g_img]).double()
    cos = float(flat @ ref / (flat.norm() * ref.norm()))
    cos_img = float(img @ ref / (img.norm() * ref.norm()))
    assert cos_img < tol["cos"] - 0.02, f"the token-attention term does not show in the gradient (cos of the image-only gradient {cos_img}): raise w_ta"
    assert cos >= tol["cos"], f"LoRA gradient with the token-attention loss: cos {cos} (image-only gradient: {cos_img})"
    ge, geo = pe_d.grad.cpu().double().reshape(-1), grads_o[-1].double().reshape(-1)
    cos = float(ge @ geo / (ge.norm() * geo.norm()))
    assert cos >= tol["cos"], f"encoder_hidden_states gradient with the token-attention loss: cos {cos}"
    return unet


@pytest.mark.parametrize("version,B", [("tinyxl", 1), ("tiny15", 2)])
def test_token_attention_loss_through_shim_cpu(version, B):
    rt = unet_mod.Runtime("cpu", B, act_dtype=torch.float32, ops=emu_ops)
    run_shim_token_attention(version, B, 16, rt, dict(loss=1e-3, cos=0.9999))


def _walk_and_install(unet, make_processor):
    """What `find_attnprocessor2_0` + `init_daam_loss` do to a diffusers UNet (ti_cross_attn_loss.py:88-112, 336-364), restated: probe the
    dotted names `<down|up>_blocks.i.attentions.j.transformer_blocks.k.attn2.processor` by getattr (AttributeError = does not exist), require the
    found object to be an `AttnProcessor2_0`, replace it with setattr on its parent and read it back."""
    from functools import reduce
    by_name = lambda root, name: reduce(getattr, name.split("."), root)
    found = []
    for kind in ("down_blocks", "up_blocks"):
        for i in range(6):
            for j in range(6):
                for k in range(12):
                    name = f"{kind}.{i}.attentions.{j}.transformer_blocks.{k}.attn2.processor"
                    try:
                        obj = by_name(unet, name)
                    except AttributeError:
                        continue
                    assert type(obj).__mro__[1].__name__ == "AttnProcessor2_0" or type(obj).__name__ == "AttnProcessor2_0", type(obj)
                    assert isinstance(obj, shim.AttnProcessor2_0)
                    found.append(name)
    installed = []
    for name in found:
        parent, attr = name.rsplit(".", 1)
        proc = make_processor(name)
        setattr(by_name(unet, parent), attr, proc)
        assert by_name(unet, name) is proc
        installed.append(proc)
    return found, installed


@pytest.mark.parametrize("version,B", [("tinyxl", 1), ("tiny15", 2)])
def test_processor_seam_installed_by_module_path_cpu(version, B):
    """VERDICT r3 'Seam 2 as the reference installs it': the reference's hook installer finds the attn2 processors by module path on the
    shim and the objects it puts there receive the score maps (in the graph) on the next call - no `keep_daam_maps` flag needed."""
    class Installed:                 # stands for DAAMLossAttnProcessor2_0(name): a foreign class with the two attributes the reference reads
        def __init__(self, name):
            self.name, self.cross_attention_scores = name, None

    rt = unet_mod.Runtime("cpu", B, act_dtype=torch.float32, ops=emu_ops)
    cfg = U.CONFIGS[version]
    sd = {k: v.to(torch.bfloat16).float() for k, v in U.init_unet_state(cfg, seed=0).items()}
    unet = shim.get_peft_model(version, sd, shim.LoraConfig(r=8, lora_alpha=8.0), batch_size=B, runtime=rt)
    unet.unet.arena.load(U.init_lora(cfg, 8, seed=1, b_std=0.03))
    unet.requires_grad_(True)
    hooked = [a.name for a in unet.unet.cross_attns if a.hooked]
    found, installed = _walk_and_install(unet, Installed)
    assert found == [n + ".processor" for n in hooked] and len(found) > 0
    assert not any(n.startswith("mid_block") for n in found)
    assert unet.down_blocks[int(found[0].split(".")[1])].attentions[0].transformer_blocks[0].attn2.processor is installed[0]      # index access too
    with pytest.raises(AttributeError):
        unet.down_blocks.no_such_child
    add = {"text_embeds": torch.zeros(B, cfg["proj_class_in"] - 6 * cfg["addition_time_embed_dim"]), "time_ids": torch.tensor([[1024., 1024, 0, 0, 128, 128]] * B)} if cfg["addition"] else None
    ehs = torch.randn(B, 77, cfg["cross_dim"], requires_grad=True)
    x, t = torch.randn(B, 4, 16, 16), torch.tensor([500] * B)
    assert unet.keep_daam_maps is False
    pred = unet(x, t, encoder_hidden_states=ehs, added_cond_kwargs=add)[0]
    with torch.no_grad():
        _, daam = U.unet_forward(cfg, sd, x, t, ehs.detach(), add, lora=dict(unet.unet.arena.export()), return_daam=True)
    assert len(daam) == len(installed)
    for proc, (name, s) in zip(installed, daam):
        assert proc.name == name + ".processor" and proc.cross_attention_scores.requires_grad
        torch.testing.assert_close(proc.cross_attention_scores.detach(), s, rtol=2e-3, atol=2e-3)
    # a loss on the installed processors' maps alone reaches the text conditioning through the module's backward
    sum(p.cross_attention_scores.float().pow(2).mean() for p in installed).backward()
    assert ehs.grad is not None and float(ehs.grad.abs().max()) > 0
    assert unet.keep_daam_maps is False
