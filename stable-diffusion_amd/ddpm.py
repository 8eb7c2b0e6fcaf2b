"""Host-side mirror of the reference's LatentDiffusion for the inference path.

Keeps the names the sampler and the scripts use (stable_diffusion.py:3666-3895, :4044-4346,
:4488-4589, :4660-4847): ``apply_model``, ``decode_first_stage``, ``q_sample``, ``get_first_stage_encoding``,
the ancestral sampler (``p_mean_variance``, ``p_sample``, ``p_sample_loop``, ``sample``), the schedule
buffers and ``num_timesteps``.  The schedule is computed on the host in float64 exactly
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

    Accepts the scripts' parameter dictionaries (txt2img.py:14-42) plus ``compute_dtype`` ("fp16" / "bf16" fast
    path, "fp32" parity mode; default: what ``unet_config['use_fp16']`` selects, i.e. fp32 for the scripts'
    dictionaries).  ``cond_stage_config`` is not
    instantiated: the text encoder lives in a third-party package and the conditioning enters this
    path as a tensor (see ``get_learned_conditioning``).
    """

    def __init__(self, first_stage_config, cond_stage_config=None, unet_config=None, linear_start=1e-4,
                 linear_end=2e-2, cosine_s=8e-3, timesteps=1000, beta_schedule="linear", scale_factor=1.0,
                 conditioning_key=None, concat_mode=True, parameterization="eps", v_posterior=0.,
                 given_betas=None, compute_dtype=None, device=None, log_every_t=100, image_size=256, channels=3,
                 **ignored):
        assert parameterization in ["eps", "x0"], 'currently only supporting "eps" and "x0"'
        if conditioning_key is None:
            conditioning_key = 'concat' if concat_mode else 'crossattn'
        if cond_stage_config == '__is_unconditional__':
            conditioning_key = None
        self.parameterization = parameterization
        self.scale_factor = scale_factor
        self.v_posterior = v_posterior
        self.log_every_t = log_every_t
        self.image_size, self.channels = image_size, channels
        self.clip_denoised = False      # LatentDiffusion overrides DDPM's default (:4089)
        self.device = torch.device(device if device is not None else "cuda")
        # not given: the reference's switch decides (unet_config's use_fp16, default False -> fp32 numerics)
        if compute_dtype is None:
            compute_dtype = "fp16" if (unet_config or {}).get("use_fp16", False) else "fp32"
        self.compute_dtype = compute_dtype
        self.model = DiffusionWrapper(unet_config, conditioning_key, compute_dtype=compute_dtype, device=self.device)
        self.first_stage_model = AutoencoderKL(**{k: v for k, v in first_stage_config.items() if k != "monitor"},
                                               compute_dtype=compute_dtype, device=self.device)
        # the text transformer is built when its weights (cond_stage_model.transformer.*) are loaded
        self.cond_stage_model = None
        self.cond_stage_config = cond_stage_config
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
        if any(k.startswith("cond_stage_model.transformer.") for k in state_dict):
            from .clip import FrozenCLIPEmbedder
            params = {}
            if isinstance(self.cond_stage_config, dict):
                params = {k: v for k, v in (self.cond_stage_config.get("params") or {}).items()
                          if k in ("version", "max_length", "tokenizer", "config")}
            self.cond_stage_model = FrozenCLIPEmbedder(device=self.device, compute_dtype=self.compute_dtype, **params)
            self.cond_stage_model.load_state_dict(state_dict, prefix="cond_stage_model.")
            known = known + ("cond_stage_model.",)
        unexpected = [k for k in state_dict if not k.startswith(known)]
        return [], unexpected

    # ---- conditioning ------------------------------------------------------------------------------
    def get_learned_conditioning(self, c):
        """``self.cond_stage_model.encode(c)`` (:4408-4420).  A floating-point tensor is taken to be the
        (B,77,context_dim) conditioning itself and returned as is; token ids (an integer tensor) or text go
        through the text transformer when its weights were loaded."""
        if torch.is_tensor(c) and c.is_floating_point():
            return c.to(self.device)
        if self.cond_stage_model is None:
            raise NotImplementedError("no text transformer weights (cond_stage_model.transformer.*) were loaded: "
                                      "pass the (B,77,768) CLIP hidden states as a tensor")
        return self.cond_stage_model.encode(c)

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
        return ops.noise_to_timesteps(x_start, noise, t, self.sqrt_alphas_cumprod, self.sqrt_one_minus_alphas_cumprod)

    # ---- ancestral (DDPM) sampling -------------------------------------------------------------------
    def _uniform_t(self, t, b):
        t = torch.as_tensor(t).reshape(-1).cpu().long()
        assert t.numel() == b, "one timestep per sample"
        return t

    def _ddpm_update(self, x, eps, t, clip_denoised, noise, temperature, want_x0):
        """Fused x0 / posterior mean / noise update, one launch per distinct timestep in ``t``
        (the sampling loops pass one value for the whole batch)."""
        b = x.shape[0]
        x_prev = torch.empty_like(x)
        x0 = torch.empty_like(x) if want_x0 else None
        # exp on the fp32 table entry, as the reference's tensor op does
        sd = (0.5 * self.posterior_log_variance_clipped).exp()
        lo = 0
        while lo < b:
            hi = lo + 1
            while hi < b and t[hi] == t[lo]:
                hi += 1
            ti = int(t[lo])
            mask = 0.0 if ti == 0 else 1.0
            xp, p0 = ops.ddpm_step(x[lo:hi], eps[lo:hi], self.sqrt_recip_alphas_cumprod[ti],
                                   self.sqrt_recipm1_alphas_cumprod[ti], self.posterior_mean_coef1[ti],
                                   self.posterior_mean_coef2[ti], mask * float(sd[ti]),
                                   noise=None if noise is None else noise[lo:hi], temperature=temperature,
                                   clip_denoised=clip_denoised, want_pred_x0=want_x0, out=x_prev[lo:hi])
            if want_x0:
                x0[lo:hi] = p0
            lo = hi
        return x_prev, x0

    def p_mean_variance(self, x, c, t, clip_denoised, return_codebook_ids=False, quantize_denoised=False,
                        return_x0=False, score_corrector=None, corrector_kwargs=None):
        """stable_diffusion.py:4660-4690: (model_mean, posterior_variance[t], posterior_log_variance[t][, x0])."""
        if return_codebook_ids or quantize_denoised:
            raise NotImplementedError("VQ first stage options are not on this path")
        x = x.to(self.device, torch.float32).contiguous()
        b = x.shape[0]
        tt = self._uniform_t(t, b)
        model_out = self.apply_model(x, torch.as_tensor(t).to(self.device), c)
        if score_corrector is not None:
            assert self.parameterization == "eps"
            model_out = score_corrector.modify_score(self, model_out, x, t, c, **(corrector_kwargs or {}))
        if self.parameterization != "eps":
            raise NotImplementedError("the fused update implements the eps parameterization")
        mean, x0 = self._ddpm_update(x, model_out.contiguous(), tt, clip_denoised, None, 1.0, return_x0)
        shape = (b,) + (1,) * (x.dim() - 1)
        var = self.posterior_variance[tt].reshape(shape).to(self.device)
        logvar = self.posterior_log_variance_clipped[tt].reshape(shape).to(self.device)
        return (mean, var, logvar, x0) if return_x0 else (mean, var, logvar)

    @torch.no_grad()
    def p_sample(self, x, c, t, clip_denoised=False, repeat_noise=False, return_codebook_ids=False,
                 quantize_denoised=False, return_x0=False, temperature=1., noise_dropout=0., score_corrector=None,
                 corrector_kwargs=None, noise=None, _eps_out=None):
        """stable_diffusion.py:4693-4721.  One UNet call and one fused update kernel; ``noise`` may be
        injected (the reference draws it with torch.randn); ``_eps_out`` names the UNet's output buffer (the
        sampling loops keep it, x and t at fixed addresses so that the UNet's launch graph is replayed)."""
        if return_codebook_ids or quantize_denoised:
            raise NotImplementedError("VQ first stage options are not on this path")
        x = x.to(self.device, torch.float32).contiguous()
        b = x.shape[0]
        tt = self._uniform_t(t, b)
        model_out = self.apply_model(x, torch.as_tensor(t).to(self.device), c, out=_eps_out)
        if score_corrector is not None:
            assert self.parameterization == "eps"
            model_out = score_corrector.modify_score(self, model_out, x, t, c, **(corrector_kwargs or {}))
        if self.parameterization != "eps":
            raise NotImplementedError("the fused update implements the eps parameterization")
        if noise is None:
            if repeat_noise:
                noise = torch.randn((1, *x.shape[1:]), device=self.device).repeat(b, *((1,) * (x.dim() - 1)))
            else:
                noise = torch.randn(x.shape, device=self.device)
        noise = noise.to(self.device, torch.float32).contiguous()
        if noise_dropout > 0.:
            noise = torch.nn.functional.dropout(noise * temperature, p=noise_dropout)
            temperature = 1.0
        x_prev, x0 = self._ddpm_update(x, model_out.contiguous(), tt, clip_denoised, noise, temperature, return_x0)
        return (x_prev, x0) if return_x0 else x_prev

    @staticmethod
    def _first_n(cond, n):
        """The leading n conditioning entries, through the dict / list nesting the scripts use (:4742-4746)."""
        if cond is None:
            return None
        cut = lambda v: [e[:n] for e in v] if isinstance(v, list) else v[:n]
        return {k: cut(v) for k, v in cond.items()} if isinstance(cond, dict) else cut(cond)

    def _ancestral_loop(self, cond, img, steps, step_fn, keep, log_every_t, mask, x0, callback, img_callback):
        """Shared body of p_sample_loop / progressive_denoising: t = steps-1 .. 0, one UNet call and one fused
        update per step (``step_fn`` returns the new latent and what to log), optional known-region blend."""
        b = img.shape[0]
        unet = self.model.diffusion_model
        # x, t and eps live at fixed addresses for the whole loop: the UNet replays its captured launch graph
        # instead of recording a new one per step
        xbuf = img.to(self.device, torch.float32).clone().contiguous()
        ts = torch.empty((b,), device=self.device, dtype=torch.long)
        ebuf = torch.empty_like(xbuf)
        if mask is not None:
            x0_d = x0.to(self.device, torch.float32).contiguous()
            m_d = mask.to(self.device, torch.float32).expand(b, -1, -1, -1).contiguous()
        for i in range(steps - 1, -1, -1):
            ts.fill_(i)
            unet.set_uniform_timesteps(True)      # one timestep for the whole batch; off again before callbacks run
            try:
                img, logged = step_fn(xbuf, cond, ts, i, ebuf)
            finally:
                unet.set_uniform_timesteps(False)
            if mask is not None:
                # img = q_sample(x0, ts) * mask + (1 - mask) * img  (:4822-4824), one kernel
                img = ops.q_sample_blend(x0_d, torch.randn_like(x0_d), m_d, img.contiguous(),
                                         self.sqrt_alphas_cumprod[i], self.sqrt_one_minus_alphas_cumprod[i])
            if i % log_every_t == 0 or i == steps - 1:
                keep.append(logged if logged is not None else img)
            if callback:
                callback(i)
            if img_callback:
                img_callback(img, i)
            xbuf.copy_(img)
        return img

    @torch.no_grad()
    def p_sample_loop(self, cond, shape, return_intermediates=False, x_T=None, verbose=True, callback=None,
                      timesteps=None, quantize_denoised=False, mask=None, x0=None, img_callback=None, start_T=None,
                      log_every_t=None):
        """stable_diffusion.py:4780-4829: ancestral sampling from x_T (or fresh noise) down to t = 0."""
        img = torch.randn(shape, device=self.device) if x_T is None else x_T.to(self.device, torch.float32)
        steps = self.num_timesteps if timesteps is None else timesteps
        if start_T is not None:
            steps = min(steps, start_T)
        if mask is not None:
            assert x0 is not None
            assert x0.shape[2:3] == mask.shape[2:3]  # spatial size has to match
        intermediates = [img]
        step = lambda im, c, ts, i, ebuf: (self.p_sample(im, c, ts, clip_denoised=self.clip_denoised,
                                                         quantize_denoised=quantize_denoised, _eps_out=ebuf), None)
        img = self._ancestral_loop(cond, img, steps, step, intermediates, log_every_t or self.log_every_t, mask, x0,
                                   callback, img_callback)
        return (img, intermediates) if return_intermediates else img

    @torch.no_grad()
    def progressive_denoising(self, cond, shape, verbose=True, callback=None, quantize_denoised=False,
                              img_callback=None, mask=None, x0=None, temperature=1., noise_dropout=0.,
                              score_corrector=None, corrector_kwargs=None, batch_size=None, x_T=None, start_T=None,
                              log_every_t=None):
        """stable_diffusion.py:4723-4777: the same loop, logging the x0 predictions; ``shape`` excludes the batch
        axis when ``batch_size`` is given."""
        if batch_size is not None:
            shape = [batch_size] + list(shape)
        b = shape[0]
        img = torch.randn(shape, device=self.device) if x_T is None else x_T.to(self.device, torch.float32)
        steps = self.num_timesteps if start_T is None else min(self.num_timesteps, start_T)
        temps = [temperature] * steps if isinstance(temperature, float) else temperature
        if mask is not None:
            assert x0 is not None
        x0_log = []
        step = lambda im, c, ts, i, ebuf: self.p_sample(im, c, ts, clip_denoised=self.clip_denoised,
                                                        quantize_denoised=quantize_denoised, return_x0=True,
                                                        temperature=temps[i], noise_dropout=noise_dropout,
                                                        score_corrector=score_corrector,
                                                        corrector_kwargs=corrector_kwargs, _eps_out=ebuf)
        img = self._ancestral_loop(self._first_n(cond, b), img, steps, step, x0_log, log_every_t or self.log_every_t,
                                   mask, x0, callback, img_callback)
        return img, x0_log

    @torch.no_grad()
    def sample(self, cond, batch_size=16, return_intermediates=False, x_T=None, verbose=True, timesteps=None,
               quantize_denoised=False, mask=None, x0=None, shape=None, **kwargs):
        """stable_diffusion.py:4832-4847."""
        if shape is None:
            shape = (batch_size, self.channels, self.image_size, self.image_size)
        return self.p_sample_loop(self._first_n(cond, batch_size), shape, return_intermediates=return_intermediates,
                                  x_T=x_T, verbose=verbose, timesteps=timesteps, quantize_denoised=quantize_denoised,
                                  mask=mask, x0=x0)
