"""Host-side mirror of the reference's LatentDiffusion for the inference path.

Keeps the names the sampler and the scripts use (stable_diffusion.py:3666-3895, :4044-4346,
:4488-4589): ``apply_model``, ``decode_first_stage``, ``q_sample``, ``get_first_stage_encoding``,
the schedule buffers and ``num_timesteps``.  The schedule is computed on the host in float64 exactly
as ``register_schedule`` does; everything that touches a latent runs on the GPU through libsdmi.
"""
import functools

import numpy as np
import torch

from . import ops
from .autoencoder import AutoencoderKL
from .unet import DiffusionWrapper


def make_beta_schedule(schedule, n_timestep, linear_start=1e-4, linear_end=2e-2, cosine_s=8e-3):
    """stable_diffusion.py:3565-3581.  NB: in this reference "linear" interpolates the betas
    themselves; the square-root interpolation other code bases call "linear" is "sqrt_linear"."""
    if schedule == "linear":
        betas = torch.linspace(linear_start, linear_end, n_timestep, dtype=torch.float64)
    elif schedule == "sqrt_linear":
        betas = torch.linspace(linear_start ** 0.5, linear_end ** 0.5, n_timestep, dtype=torch.float64) ** 2
    elif schedule == "sqrt":
        betas = torch.linspace(linear_start, linear_end, n_timestep, dtype=torch.float64) ** 0.5
    elif schedule == "cosine":
        timesteps = torch.arange(n_timestep + 1, dtype=torch.float64) / n_timestep + cosine_s
        alphas = torch.cos(timesteps / (1 + cosine_s) * np.pi / 2).pow(2)
        alphas = alphas / alphas[0]
        betas = torch.clip(1 - alphas[1:] / alphas[:-1], min=0, max=0.999)
    else:
        raise ValueError(f"schedule '{schedule}' unknown.")
    return betas.numpy()


class LatentDiffusion:
    """Inference-side stand-in for stable_diffusion.LatentDiffusion.

    Accepts the scripts' parameter dictionaries (txt2img.py:14-42).  ``cond_stage_config`` is not
    instantiated: the text encoder lives in a third-party package and the conditioning enters this
    path as a tensor (see ``get_learned_conditioning``).
    """

    def __init__(self, first_stage_config, cond_stage_config=None, unet_config=None, linear_start=1e-4,
                 linear_end=2e-2, cosine_s=8e-3, timesteps=1000, beta_schedule="linear", scale_factor=1.0,
                 conditioning_key=None, concat_mode=True, parameterization="eps", v_posterior=0.,
                 given_betas=None, compute_dtype="fp16", device=None, **ignored):
        assert parameterization in ["eps", "x0"], 'currently only supporting "eps" and "x0"'
        if conditioning_key is None:
            conditioning_key = 'concat' if concat_mode else 'crossattn'
        if cond_stage_config == '__is_unconditional__':
            conditioning_key = None
        self.parameterization = parameterization
        self.scale_factor = scale_factor
        self.v_posterior = v_posterior
        self.device = torch.device(device if device is not None else "cuda")
        self.compute_dtype = compute_dtype
        self.model = DiffusionWrapper(unet_config, conditioning_key, compute_dtype=compute_dtype, device=self.device)
        self.first_stage_model = AutoencoderKL(**{k: v for k, v in first_stage_config.items() if k != "monitor"},
                                               compute_dtype=compute_dtype, device=self.device)
        self.cond_stage_model = None
        self.register_schedule(given_betas=given_betas, beta_schedule=beta_schedule, timesteps=timesteps,
                               linear_start=linear_start, linear_end=linear_end, cosine_s=cosine_s)

    def eval(self):
        return self

    def cpu(self):
        raise RuntimeError("this model runs on the MI355X only (no CPU path)")

    # ---- schedule (stable_diffusion.py:3755-3793) ------------------------------------------------
    def register_schedule(self, given_betas=None, beta_schedule="linear", timesteps=1000, linear_start=1e-4,
                          linear_end=2e-2, cosine_s=8e-3):
        betas = given_betas if given_betas is not None else make_beta_schedule(
            beta_schedule, timesteps, linear_start=linear_start, linear_end=linear_end, cosine_s=cosine_s)
        alphas = 1. - betas
        alphas_cumprod = np.cumprod(alphas, axis=0)
        alphas_cumprod_prev = np.append(1., alphas_cumprod[:-1])
        self.num_timesteps = int(betas.shape[0])
        self.linear_start, self.linear_end = linear_start, linear_end
        to_torch = functools.partial(torch.tensor, dtype=torch.float32)
        self.betas = to_torch(betas)
        self.alphas_cumprod = to_torch(alphas_cumprod)
        self.alphas_cumprod_prev = to_torch(alphas_cumprod_prev)
        self.sqrt_alphas_cumprod = to_torch(np.sqrt(alphas_cumprod))
        self.sqrt_one_minus_alphas_cumprod = to_torch(np.sqrt(1. - alphas_cumprod))
        self.log_one_minus_alphas_cumprod = to_torch(np.log(1. - alphas_cumprod))
        self.sqrt_recip_alphas_cumprod = to_torch(np.sqrt(1. / alphas_cumprod))
        self.sqrt_recipm1_alphas_cumprod = to_torch(np.sqrt(1. / alphas_cumprod - 1))
        posterior_variance = (1 - self.v_posterior) * betas * (1. - alphas_cumprod_prev) / (1. - alphas_cumprod) + \
            self.v_posterior * betas
        self.posterior_variance = to_torch(posterior_variance)
        self.posterior_log_variance_clipped = to_torch(np.log(np.maximum(posterior_variance, 1e-20)))
        self.posterior_mean_coef1 = to_torch(betas * np.sqrt(alphas_cumprod_prev) / (1. - alphas_cumprod))
        # as in the reference (:3792-3793): multiplied by (1 - alphas_cumprod), not divided
        self.posterior_mean_coef2 = to_torch(np.sqrt(alphas) * (1. - alphas_cumprod_prev) * (1. - alphas_cumprod))

    # ---- weights ---------------------------------------------------------------------------------
    def load_state_dict(self, state_dict, strict=False):
        """Full-checkpoint key layout: ``model.diffusion_model.*`` and ``first_stage_model.*``."""
        self.model.diffusion_model.load_state_dict(state_dict, strict=False, prefix="model.diffusion_model.")
        self.first_stage_model.load_state_dict(state_dict, strict=False, prefix="first_stage_model.")
        known = ("model.diffusion_model.", "first_stage_model.")
        unexpected = [k for k in state_dict if not k.startswith(known)]
        return [], unexpected

    # ---- conditioning ------------------------------------------------------------------------------
    def get_learned_conditioning(self, c):
        """The reference runs FrozenCLIPEmbedder here (:4408); on this path the conditioning is already
        a (B,77,context_dim) tensor and is returned as is."""
        if torch.is_tensor(c):
            return c.to(self.device)
        raise NotImplementedError("text encoding is outside the MI355X denoising path: pass the "
                                  "(B,77,768) CLIP hidden states as a tensor")

    # ---- denoiser --------------------------------------------------------------------------------
    def apply_model(self, x_noisy, t, cond, return_ids=False, out=None):
        """stable_diffusion.py:4488-4497, :4584-4589 (non-split branch)."""
        if isinstance(cond, dict):
            pass
        else:
            if not isinstance(cond, list):
                cond = [cond]
            key = 'c_concat' if self.model.conditioning_key == 'concat' else 'c_crossattn'
            cond = {key: cond}
        x_recon = self.model(x_noisy, t, out=out, **cond)
        if isinstance(x_recon, tuple) and not return_ids:
            return x_recon[0]
        return x_recon

    # ---- first stage -------------------------------------------------------------------------------
    def decode_first_stage(self, z, predict_cids=False, force_not_quantize=False):
        """z = 1/scale_factor * z; AutoencoderKL.decode(z)   (:4300, :4346)."""
        if predict_cids:
            raise NotImplementedError("predict_cids (VQ first stage) is not on this path")
        return self.first_stage_model.decode(z, _z_scale=1. / self.scale_factor)

    def encode_first_stage(self, x):
        return self.first_stage_model.encode(x)

    def get_first_stage_encoding(self, encoder_posterior, noise=None):
        """scale_factor * posterior.sample() or scale_factor * tensor (:4264-4271); ``noise`` may be injected."""
        from .autoencoder import DiagonalGaussianDistribution
        if isinstance(encoder_posterior, DiagonalGaussianDistribution):
            return encoder_posterior.sample(noise=noise, _scale=self.scale_factor)
        if isinstance(encoder_posterior, torch.Tensor):
            return self.scale_factor * encoder_posterior
        raise NotImplementedError(f"encoder_posterior of type '{type(encoder_posterior)}' not yet implemented")

    # ---- forward diffusion ---------------------------------------------------------------------------
    def q_sample(self, x_start, t, noise=None):
        """sqrt(acp[t]) * x_start + sqrt(1-acp[t]) * noise   (:3892-3895), per-sample t."""
        x_start = x_start.to(self.device, torch.float32).contiguous()
        if noise is None:
            noise = torch.randn_like(x_start)
        noise = noise.to(self.device, torch.float32).contiguous()
        t = torch.as_tensor(t).cpu().long()
        out = torch.empty_like(x_start)
        for i in range(x_start.shape[0]):
            out[i] = ops.axpby(x_start[i], noise[i], float(self.sqrt_alphas_cumprod[t[i]]),
                               float(self.sqrt_one_minus_alphas_cumprod[t[i]]))
        return out
