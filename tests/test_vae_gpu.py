"""VAE decode / encode on the MI355X (bf16 HIP plans) against the fp32 oracle restatement; tolerances: bf16 activations
through ~30 conv/norm layers -> max-abs error <= 5e-2 of the output range, cosine >= 0.995."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _cmp(a, b, what):
    a, b = a.float().cpu(), b.float().cpu()
    err = float((a - b).abs().max()) / float(b.abs().max())
    cos = float((a.reshape(-1).double() @ b.reshape(-1).double()) / (a.double().norm() * b.double().norm()))
    assert err <= 5e-2 and cos >= 0.995, (what, err, cos)


@pytest.mark.parametrize("kind,B,h,w", [("tiny", 1, 8, 8), ("tiny", 2, 16, 8), ("sd", 1, 32, 32)])
def test_vae_gpu_matches_oracle(kind, B, h, w):
    """tiny: the toy autoencoder; sd: the REAL AutoencoderKL topology of SD1.5 / SDXL (83.7 M parameters, random-init) on a 32 x 32 latent / 256 px image."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from oracle import vae_ref as V
    from sd_lora_trainer_amd import vae
    import sd_lora_trainer_amd.unet as M
    cfg = V.CONFIGS[kind]
    sd = {k: v.to(torch.bfloat16).float() for k, v in V.init_state(cfg, seed=0).items()}
    g = torch.Generator().manual_seed(1)
    z = torch.randn(B, 4, h, w, generator=g)
    rt = M.Runtime("cuda:0", B)
    dec, enc = vae.VaeDecoder(rt, sd), vae.VaeEncoder(rt, sd)
    img = dec.decode(z.cuda())
    _cmp(img, V.decode(cfg, sd, z), "decode")
    f = 2 ** (len(cfg["block_out_channels"]) - 1)
    x = torch.tanh(torch.randn(B, 3, f * h, f * w, generator=g))
    _cmp(enc.encode_moments(x.cuda()), V.encode_moments(cfg, sd, x), "encode moments")
    # render path end to end: latents (scaled) -> decode(latents / scaling_factor) -> [0, 1] image
    out = vae.postprocess(dec.decode(z.cuda() * 0.13025 / 0.13025))
    assert float(out.min()) >= 0.0 and float(out.max()) <= 1.0
