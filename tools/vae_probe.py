"""Times the full-size VAE (SD/SDXL configuration, random weights): decode of a 128x128 latent (1024 px render) and encode of
a 1024 px image.  python tools/vae_probe.py"""
import sys
import time

import torch

sys.path.insert(0, ".")
from oracle import vae_ref as V          # noqa: E402  (weights generator only)
from sd_lora_trainer_amd import vae      # noqa: E402
import sd_lora_trainer_amd.unet as M     # noqa: E402

cfg = V.CONFIGS["sd"]
sd = {k: v.cuda() for k, v in V.init_state(cfg, seed=0).items()}
rt = M.Runtime("cuda:0", 1)
dec, enc = vae.VaeDecoder(rt, sd), vae.VaeEncoder(rt, sd)
z = torch.randn(1, 4, 128, 128, device="cuda")
img = torch.tanh(torch.randn(1, 3, 1024, 1024, device="cuda"))
for name, fn in (("decode 128x128 latent -> 1024 px", lambda: dec.decode(z)), ("encode 1024 px -> moments", lambda: enc.encode_moments(img))):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        out = fn()
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) / 3 * 1e3:.1f} ms, out {tuple(out.shape)}, finite {bool(torch.isfinite(out).all())}, "
          f"mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
