"""sdmi -- MI355X-native Stable-Diffusion-v1 denoising hot path.

Python mirror of the reference's inference-side classes (UNetModel, DiffusionWrapper,
LatentDiffusion, DDIMSampler, AutoencoderKL, FrozenCLIPEmbedder) over hand-written gfx950 kernels in libsdmi.so.
Importing the package needs neither the library nor a GPU; using a model does.
"""
from .structure import (sd_v1_unet_config, sd_v1_ddconfig, sd_v1_model_params, unet_structure,
                        unet_param_shapes, vae_decoder_structure, vae_decoder_param_shapes,
                        vae_encoder_structure, vae_encoder_param_shapes)
from . import synth


def __getattr__(name):
    # device-side classes are imported lazily so that CPU-only tooling can import the package
    if name in ("UNetModel", "DiffusionWrapper"):
        from . import unet
        return getattr(unet, name)
    if name in ("LatentDiffusion", "make_beta_schedule"):
        from . import ddpm
        return getattr(ddpm, name)
    if name in ("DDIMSampler", "make_ddim_timesteps", "make_ddim_sampling_parameters"):
        from . import ddim
        return getattr(ddim, name)
    if name in ("AutoencoderKL", "DiagonalGaussianDistribution"):
        from . import autoencoder
        return getattr(autoencoder, name)
    if name in ("FrozenCLIPEmbedder", "clip_text_param_shapes", "CLIP_VIT_L14_TEXT"):
        from . import clip
        return getattr(clip, name)
    raise AttributeError(name)
