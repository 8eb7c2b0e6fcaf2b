"""Import the reference's stable_diffusion.py in the authoring container (never on the GPU box).

The reference's module top imports four packages that are not installed here and that the
denoising path never touches (pytorch_lightning, torchvision, kornia, clip); they are replaced by
empty stand-in modules so that the *unmodified* reference file can be imported and its UNetModel /
DDIMSampler / AutoencoderKL / LatentDiffusion classes executed on CPU to produce golden vectors.
Used only by ``oracle/make_golden.py``.
"""
import sys
import types

REFERENCE_DIR = "/root/reference"


def import_reference():
    if "stable_diffusion" in sys.modules:
        return sys.modules["stable_diffusion"]
    import torch
    import torch.nn as nn
    # transformers probes torchvision's availability at import time; import it before the stand-in
    from transformers import CLIPTokenizer, CLIPTextModel  # noqa: F401

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class LightningModule(nn.Module):
        @property
        def device(self):
            try:
                return next(self.parameters()).device
            except StopIteration:
                return torch.device("cpu")

    def seed_everything(seed):
        torch.manual_seed(seed)

    pl = mod("pytorch_lightning", LightningModule=LightningModule, seed_everything=seed_everything)
    pl.utilities = mod("pytorch_lightning.utilities", rank_zero_only=lambda f: f)
    tv = mod("torchvision")
    tv.models = mod("torchvision.models")
    tv.utils = mod("torchvision.utils", make_grid=lambda *a, **k: None)
    mod("kornia")
    mod("clip")
    sys.path.insert(0, REFERENCE_DIR)
    try:
        import stable_diffusion
    finally:
        sys.path.remove(REFERENCE_DIR)
    return stable_diffusion
