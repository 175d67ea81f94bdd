"""The step's input stage downstream of the VAE encoder: the reference's `PreprocessedDataset`
(/root/reference trainer/dataset.py:31-193) keeps, per image, the VAE POSTERIOR (not a latent) and the latent-
resolution mask, and draws a fresh `latent_dist.sample() * scaling_factor` on every fetch (dataset.py:184-193) - the noise
of the encoder is part of the training signal.  Captions are lower-cased and the trigger words substituted once
(dataset.py:46-52).  `LatentCache` holds the encoder's moments tensor `[1, 8, h, w]` (mean | logvar) per image - what
`vae.encode(x).latent_dist.parameters` holds; `LatentCache.from_folder` produces them with vae.VaeEncoder (once per job).

`DiagonalGaussian` restates diffusers' DiagonalGaussianDistribution (0.29.2, third party): logvar clamped to [-30, 20],
std = exp(0.5 logvar), sample = mean + std * N(0, 1).
"""
import numpy as np
import torch


class DiagonalGaussian:
    def __init__(self, parameters):
        self.parameters = parameters
        self.mean, logvar = torch.chunk(parameters, 2, dim=1)
        self.logvar = torch.clamp(logvar, -30.0, 20.0)
        self.std = torch.exp(0.5 * self.logvar)

    def sample(self, generator=None):
        noise = torch.randn(self.mean.shape, generator=generator, device=self.mean.device, dtype=self.mean.dtype)
        return self.mean + self.std * noise

    def mode(self):
        return self.mean


def prepare_image(pil_image, w=512, h=512):
    """dataset.py:11-17: bicubic resize to the training size, then diffusers' VaeImageProcessor.preprocess for a PIL input
    whose size is already a multiple of 8: RGB / 255 -> [1, 3, h, w] -> 2x - 1 (third party, restated)."""
    from PIL import Image
    pil_image = pil_image.resize((w, h), resample=Image.BICUBIC, reducing_gap=1)
    arr = np.array(pil_image.convert("RGB")).astype(np.float32) / 255.0
    return torch.from_numpy(arr).permute(2, 0, 1).unsqueeze(0) * 2.0 - 1.0


def prepare_mask(pil_image, w=512, h=512):
    """dataset.py:19-28: bicubic resize to the training size, luminance / 255 -> [1, 1, h, w] fp32."""
    from PIL import Image
    pil_image = pil_image.resize((w, h), resample=Image.BICUBIC, reducing_gap=1)
    arr = np.array(pil_image.convert("L")).astype(np.float32) / 255.0
    return torch.from_numpy(np.expand_dims(arr, 0)).unsqueeze(0)


def latent_mask(mask_image, size, latent_hw, channels=4):
    """dataset.py:162-175: the image-resolution mask, nearest-resized to the latent grid and repeated over the latent
    channels; `None` (no mask_path column) -> all ones (dataset.py:159-160).  -> [channels, h, w]."""
    if mask_image is None:
        return torch.ones(channels, *latent_hw)
    m = prepare_mask(mask_image, size[0], size[1]).float()
    m = torch.nn.functional.interpolate(m, size=tuple(latent_hw), mode="nearest")
    return m.repeat(1, channels, 1, 1).squeeze()


def process_captions(captions, substitute_caption_map=None):
    """dataset.py:46-52: lower-case, substitute (keys lower-cased too, plain substring replacement), NaN -> ''."""
    out = []
    for c in captions:
        if c is None or (isinstance(c, float) and c != c):
            out.append("")
            continue
        c = str(c).lower()
        for key, value in (substitute_caption_map or {}).items():
            c = c.replace(key.lower(), value)
        out.append(c)
    return out


class LatentCache:
    """In-memory variant of PreprocessedDataset (`do_cache`, < 500 images, dataset.py:66-77)."""

    def __init__(self, posterior_params, mask_images, captions, *, scaling_factor, size, substitute_caption_map=None):
        """posterior_params: list of [1, 8, h, w] moments from the VAE encoder; mask_images: list of PIL images or None."""
        self.dists = [DiagonalGaussian(p) for p in posterior_params]
        hw = tuple(self.dists[0].mean.shape[-2:])
        self.masks = [latent_mask(None if mask_images is None else mask_images[i], size, tuple(d.mean.shape[-2:]), d.mean.shape[1])
                      for i, d in enumerate(self.dists)]
        self.captions = process_captions(captions, substitute_caption_map)
        self.scaling_factor, self.latent_hw = scaling_factor, hw

    @classmethod
    def from_folder(cls, data_dir, encoder, *, size, scaling_factor, substitute_caption_map=None):
        """PreprocessedDataset.__init__ (dataset.py:31-90): `captions.csv` (image_path, caption[, mask_path]) in data_dir,
        every image encoded ONCE by the VAE encoder (vae.VaeEncoder.encode_moments -> the posterior's moments)."""
        import csv
        import os
        from PIL import Image
        with open(os.path.join(data_dir, "captions.csv"), newline="") as fh:
            rows = list(csv.DictReader(fh))
        posts, masks = [], []
        for r in rows:
            img = prepare_image(Image.open(os.path.join(data_dir, r["image_path"])).convert("RGB"), size[0], size[1])
            posts.append(encoder.encode_moments(img).float().cpu())
            if "mask_path" in r:
                masks.append(Image.open(os.path.join(data_dir, r["mask_path"])))
        caps = [r.get("caption") if r.get("caption") not in (None, "") else float("nan") for r in rows]
        return cls(posts, masks if masks else None, caps, scaling_factor=scaling_factor, size=size, substitute_caption_map=substitute_caption_map)

    def __len__(self):
        return len(self.dists)

    def __getitem__(self, idx, generator=None):
        """-> (caption, latent [4, h, w] freshly sampled, mask [4, h, w])   (dataset.py:184-187)."""
        latent = self.dists[idx].sample(generator) * self.scaling_factor
        return self.captions[idx], latent.squeeze().detach(), self.masks[idx].detach()

    def batch(self, indices, generator=None):
        items = [self.__getitem__(int(i), generator) for i in indices]
        return [c for c, _, _ in items], torch.stack([l for _, l, _ in items]), torch.stack([m for _, _, m in items])
