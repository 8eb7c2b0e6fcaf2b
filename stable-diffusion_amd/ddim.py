"""DDIMSampler with the reference's interface (stable_diffusion.py:5141-5412) on top of libsdmi.

The schedule tables are computed on the host with the same numpy/torch mixed-precision arithmetic as
the reference; each step is one UNet launch sequence (a HIP-graph replay once pointers repeat) and
one fused classifier-free-guidance + DDIM-update kernel.  Buffers are allocated once per call so
device pointers stay stable across the S steps.

Conditioning contract (the one call sequence of the reference that runs): ``conditioning`` is
``cat([uncond, cond])`` on the batch axis (uncond first, matching ``.chunk(2)`` at :5346) and
``unconditional_conditioning`` is merely truthy.
"""
import numpy as np
import torch
from tqdm import tqdm

from . import ops


def make_ddim_timesteps(ddim_discr_method, num_ddim_timesteps, num_ddpm_timesteps, verbose=True):
    """stable_diffusion.py:5141-5153: every (T // S)-th step ("uniform") or quadratically spaced steps up to
    0.8 T ("quad"), shifted by one so that the last entry reaches the end of the training schedule."""
    T, S = num_ddpm_timesteps, num_ddim_timesteps
    if ddim_discr_method == "uniform":
        base = np.arange(0, T, T // S, dtype=np.int64)
    elif ddim_discr_method == "quad":
        base = np.square(np.linspace(0, np.sqrt(0.8 * T), S)).astype(int)
    else:
        raise NotImplementedError(f'There is no ddim discretization method called "{ddim_discr_method}"')
    steps_out = base + 1
    if verbose:
        print(f"Selected timesteps for ddim sampler: {steps_out}")
    return steps_out


def make_ddim_sampling_parameters(alphacums, ddim_timesteps, eta, verbose=True):
    """stable_diffusion.py:5155-5165; ``alphacums`` is the fp32 CPU tensor, so ``alphas`` stays an fp32
    tensor, ``alphas_prev`` a float64 ndarray of fp32 values and ``sigmas`` a float64 tensor in which
    1/(1-alphas) was rounded to fp32 (ndarray / tensor runs as reciprocal(tensor) * ndarray)."""
    alphas = alphacums[ddim_timesteps]
    alphas_prev = np.asarray([float(alphacums[0])] + alphacums[ddim_timesteps[:-1]].tolist())
    recip = torch.reciprocal(1 - alphas).double()
    sigmas = eta * torch.sqrt(recip * torch.from_numpy(1 - alphas_prev) * (1 - alphas.double() / torch.from_numpy(alphas_prev)))
    if verbose:
        print(f"Selected alphas for ddim sampler: a_t: {alphas}; a_(t-1): {alphas_prev}")
        print(f"For the chosen value of eta, which is {eta}, "
              f"this results in the following sigma_t schedule for ddim sampler {sigmas}")
    return sigmas, alphas, alphas_prev


class DDIMSampler(object):
    def __init__(self, model, schedule="linear", **kwargs):
        super().__init__()
        self.model = model
        self.ddpm_num_timesteps = model.num_timesteps
        self.schedule = schedule

    def register_buffer(self, name, attr):
        setattr(self, name, attr)

    # derived 1000-step tables the sampler exposes: name -> function of the fp32 alphas_cumprod (as float64-free
    # numpy-on-fp32 arithmetic, the way the reference derives them at :5197-5203)
    _DERIVED = (("sqrt_alphas_cumprod", lambda ac: np.sqrt(ac)),
                ("sqrt_one_minus_alphas_cumprod", lambda ac: np.sqrt(1. - ac)),
                ("log_one_minus_alphas_cumprod", lambda ac: np.log(1. - ac)),
                ("sqrt_recip_alphas_cumprod", lambda ac: np.sqrt(1. / ac)),
                ("sqrt_recipm1_alphas_cumprod", lambda ac: np.sqrt(1. / ac - 1)))

    def make_schedule(self, ddim_num_steps, ddim_discretize="uniform", ddim_eta=0., verbose=True):
        """stable_diffusion.py:5180-5216.  All tables are small host tensors (the per-step scalars travel to the
        fused update kernel as arguments); dtypes follow the reference: fp32 copies of the model's buffers, an
        fp32 ``ddim_alphas``, float64 ``ddim_alphas_prev`` / ``ddim_sigmas``."""
        model = self.model
        acp = model.alphas_cumprod.cpu()
        assert acp.shape[0] == self.ddpm_num_timesteps, 'alphas have to be defined for each timestep'
        self.ddim_timesteps = make_ddim_timesteps(ddim_discretize, ddim_num_steps, self.ddpm_num_timesteps,
                                                  verbose=verbose)
        tables = {name: getattr(model, name) for name in ("betas", "alphas_cumprod", "alphas_cumprod_prev")}
        tables.update((name, fn(acp.numpy())) for name, fn in self._DERIVED)
        for name, value in tables.items():
            self.register_buffer(name, torch.as_tensor(value).detach().to(torch.float32).cpu().clone())
        sig, a, a_prev = make_ddim_sampling_parameters(alphacums=acp, ddim_timesteps=self.ddim_timesteps,
                                                       eta=ddim_eta, verbose=verbose)
        for name, value in (("ddim_sigmas", sig), ("ddim_alphas", a), ("ddim_alphas_prev", a_prev),
                            ("ddim_sqrt_one_minus_alphas", torch.from_numpy(np.sqrt(1. - a.numpy())))):
            self.register_buffer(name, value)
        # sigma table for sampling on the original 1000 steps (use_original_steps=True), all fp32
        a_t, a_tm1 = self.alphas_cumprod, self.alphas_cumprod_prev
        self.register_buffer('ddim_sigmas_for_original_num_steps',
                             ddim_eta * torch.sqrt((1 - a_tm1) / (1 - a_t) * (1 - a_t / a_tm1)))

    @torch.no_grad()
    def sample(self, S, batch_size, shape, conditioning=None, callback=None, normals_sequence=None,
               img_callback=None, quantize_x0=False, eta=0., mask=None, x0=None, temperature=1.,
               noise_dropout=0., score_corrector=None, corrector_kwargs=None, verbose=True, x_T=None,
               log_every_t=100, unconditional_guidance_scale=1., unconditional_conditioning=None, **kwargs):
        self.make_schedule(ddim_num_steps=S, ddim_eta=eta, verbose=verbose)
        C, H, W = shape
        size = (batch_size, C, H, W)
        if verbose:
            print(f'Data shape for DDIM sampling is {size}, eta {eta}')
        return self.ddim_sampling(conditioning, size, callback=callback, img_callback=img_callback,
                                  quantize_denoised=quantize_x0, mask=mask, x0=x0, ddim_use_original_steps=False,
                                  noise_dropout=noise_dropout, temperature=temperature,
                                  score_corrector=score_corrector, corrector_kwargs=corrector_kwargs, x_T=x_T,
                                  log_every_t=log_every_t,
                                  unconditional_guidance_scale=unconditional_guidance_scale,
                                  unconditional_conditioning=unconditional_conditioning, verbose=verbose,
                                  noise_fn=kwargs.get("noise_fn"), q_noise_fn=kwargs.get("q_noise_fn"))

    @torch.no_grad()
    def ddim_sampling(self, conditioning, shape, x_T=None, ddim_use_original_steps=False, callback=None,
                      timesteps=None, quantize_denoised=False, mask=None, x0=None, img_callback=None,
                      log_every_t=100, temperature=1., noise_dropout=0., score_corrector=None,
                      corrector_kwargs=None, unconditional_guidance_scale=1., unconditional_conditioning=None,
                      verbose=True, noise_fn=None, q_noise_fn=None):
        device = self.model.device
        b = shape[0]
        img = torch.randn(shape, device=device) if x_T is None else x_T.to(device, torch.float32)
        if timesteps is None:
            timesteps = self.ddpm_num_timesteps if ddim_use_original_steps else self.ddim_timesteps
        elif timesteps is not None and not ddim_use_original_steps:
            subset_end = int(min(timesteps / self.ddim_timesteps.shape[0], 1) * self.ddim_timesteps.shape[0]) - 1
            timesteps = self.ddim_timesteps[:subset_end]
        u_cond = unconditional_conditioning is not None and unconditional_guidance_scale != 1.
        cond = conditioning
        intermediates = {'x_inter': [img], 'pred_x0': [img]}
        time_range = reversed(range(0, timesteps)) if ddim_use_original_steps else np.flip(timesteps)
        total_steps = timesteps if ddim_use_original_steps else timesteps.shape[0]
        if verbose:
            print(f"Running DDIM Sampling with {total_steps} timesteps")
        iterator = tqdm(time_range, desc='DDIM Sampler', total=total_steps, disable=not verbose)
        st = self._state_for(img, u_cond)
        for i, step in enumerate(iterator):
            index = total_steps - i - 1
            if mask is not None:
                # img = q_sample(x0, ts) * mask + (1 - mask) * img  (:5316-5319) as one kernel; the noise of
                # q_sample (:3893) is drawn on the device unless the caller injects it
                assert x0 is not None
                if i == 0:
                    x0_d = x0.to(device, torch.float32).contiguous()
                    m_d = mask.to(device, torch.float32).expand(b, -1, -1, -1).contiguous()
                nz = q_noise_fn(x0_d.shape) if q_noise_fn is not None else torch.randn_like(x0_d)
                nz = nz.to(device, torch.float32).contiguous()
                st.set_x(ops.q_sample_blend(x0_d, nz, m_d, st.x().contiguous(),
                                            self.model.sqrt_alphas_cumprod[int(step)],
                                            self.model.sqrt_one_minus_alphas_cumprod[int(step)]))
            pred_x0 = st.step(cond, int(step), index, use_original_steps=ddim_use_original_steps,
                              quantize_denoised=quantize_denoised, temperature=temperature,
                              noise_dropout=noise_dropout, score_corrector=score_corrector,
                              corrector_kwargs=corrector_kwargs,
                              unconditional_guidance_scale=unconditional_guidance_scale, noise_fn=noise_fn,
                              want_pred_x0=img_callback is not None)
            if callback:
                callback(i)
            if img_callback:
                img_callback(pred_x0, i)
        return st.x().clone(), intermediates

    def _state_for(self, img, doubled):
        """Loop buffers are kept per (shape, doubled) so that repeated calls present the same device
        addresses to the UNet (its captured launch graph is then replayed instead of re-recorded)."""
        cache = self.__dict__.setdefault("_states", {})
        key = (tuple(img.shape), bool(doubled), str(self.model.device))
        st = cache.get(key)
        if st is None:
            st = cache[key] = self._StepState(self, img, doubled)
        else:
            st.set_x(img.to(self.model.device, torch.float32))
            st.t_val = None
        return st

    class _StepState:
        """Persistent device buffers for a sampling loop: the (possibly doubled) UNet input, the
        timestep vector and the eps output keep their addresses from step to step."""

        def __init__(self, sampler, img, doubled):
            self.s = sampler
            self.b = img.shape[0]
            self.doubled = doubled
            dev = sampler.model.device
            reps = 2 if doubled else 1
            self.x_in = img.to(dev, torch.float32).repeat(reps, 1, 1, 1).contiguous()
            self.t_in = torch.zeros((reps * self.b,), dtype=torch.int64, device=dev)
            self.eps = torch.empty_like(self.x_in)
            self.x_next = torch.empty_like(self.x_in[:self.b])
            self.pred = torch.empty_like(self.x_next)
            self.t_val = None

        def x(self):
            return self.x_in[:self.b]

        def set_x(self, x):
            self.x_in.view(-1, self.b, *self.x_in.shape[1:])[:] = x

        def step(self, cond, t_value, index, use_original_steps=False, quantize_denoised=False, temperature=1.,
                 noise_dropout=0., score_corrector=None, corrector_kwargs=None, unconditional_guidance_scale=1.,
                 noise_fn=None, want_pred_x0=True):
            """One p_sample_ddim (:5338-5375).  ``t_value`` is the step's timestep for the whole batch (what the
            sampling loops pass, :5314) or a (b,) tensor of per-sample timesteps."""
            s = self.s
            if quantize_denoised:
                raise NotImplementedError("quantize_denoised needs a VQ first stage, which is not on the MI355X "
                                          "sampling path (SD v1 uses the KL autoencoder)")
            uniform = not torch.is_tensor(t_value)
            if not uniform:
                # t_in = cat([t] * 2) (:5345); the UNet then embeds every row's own timestep
                self.t_in.view(-1, self.b)[:] = t_value.to(self.t_in.device, torch.int64).reshape(-1)
                self.t_val = None
            elif self.t_val != t_value:
                self.t_in.fill_(t_value)
                self.t_val = t_value
            # one value for every sample: the UNet may run its time-embedding MLP on a single row
            unet = s.model.model.diffusion_model
            unet.set_uniform_timesteps(uniform)
            unet.set_cfg_duplicate(self.doubled)       # x_in = cat([x] * 2): halves differ only in context
            try:
                s.model.apply_model(self.x_in, self.t_in, cond, out=self.eps)
            finally:
                unet.set_uniform_timesteps(False)
                unet.set_cfg_duplicate(False)
            eps, combine = self.eps, self.doubled
            if score_corrector is not None:
                # the corrector sees the guided score (:5347, :5351-5353): combine first, update without guidance
                assert s.model.parameterization == "eps"
                if self.doubled:
                    eps = ops.cfg_combine(self.eps, unconditional_guidance_scale)
                eps = score_corrector.modify_score(s.model, eps, self.x(), self.t_in[:self.b], cond,
                                                   **(corrector_kwargs or {}))
                eps = eps.to(self.x_next.device, torch.float32).contiguous()
                combine = False
            a_t, a_prev, sigma_t, sqrt_1m = s._scalars(index, use_original_steps)
            noise = None
            if sigma_t != 0.:
                shape = self.x_next.shape
                noise = noise_fn(shape) if noise_fn is not None else torch.randn(shape, device=self.x_next.device)
                noise = noise.to(self.x_next.device, torch.float32).contiguous()
                if noise_dropout > 0.:
                    # dropout(sigma * noise * temperature) (:5371-5373): the mask and its 1 / (1 - p) act on the noise
                    # term, whose other factors are scalars
                    noise = torch.nn.functional.dropout(noise * temperature, p=noise_dropout)
                    temperature = 1.
            from ._lib import lib, check, ptr, stream_ptr
            check(lib().sdmi_ddim_step(ptr(self.x_in), ptr(eps), 1 if combine else 0,
                                       float(unconditional_guidance_scale), a_t, a_prev, sigma_t, sqrt_1m, ptr(noise),
                                       float(temperature), ptr(self.x_next), ptr(self.pred) if want_pred_x0 else None,
                                       self.x_next.numel(), stream_ptr()), "ddim_step")
            self.set_x(self.x_next)
            return self.pred if want_pred_x0 else None

    def _scalars(self, index, use_original_steps):
        """The four per-step scalars of p_sample_ddim (:5355-5363), as python floats of fp32 values."""
        if use_original_steps:
            alphas, alphas_prev = self.model.alphas_cumprod, self.model.alphas_cumprod_prev
            sqrt_1m, sigmas = self.model.sqrt_one_minus_alphas_cumprod, self.ddim_sigmas_for_original_num_steps
        else:
            alphas, alphas_prev = self.ddim_alphas, self.ddim_alphas_prev
            sqrt_1m, sigmas = self.ddim_sqrt_one_minus_alphas, self.ddim_sigmas
        f32 = lambda v: float(np.float32(float(v)))      # torch.full(..., value) materialises fp32
        return f32(alphas[index]), f32(alphas_prev[index]), f32(sigmas[index]), f32(sqrt_1m[index])

    @torch.no_grad()
    def p_sample_ddim(self, x, c, t, index, repeat_noise=False, use_original_steps=False, quantize_denoised=False,
                      temperature=1., noise_dropout=0., score_corrector=None, corrector_kwargs=None,
                      unconditional_guidance_scale=1., unconditional_conditioning=True, noise=None):
        """One step with the reference's signature; returns (x_prev, pred_x0)."""
        doubled = bool(unconditional_conditioning) and unconditional_guidance_scale != 1.
        st = self._StepState(self, x, doubled)
        t = torch.as_tensor(t).reshape(-1)
        # one timestep for the batch (what the sampling loops pass, :5314) takes the single-row time embedding;
        # per-sample timesteps go to the UNet as they are (:5344-5346)
        t_value = int(t[0]) if bool((t == t[0]).all()) else t
        pred = st.step(c, t_value, index, use_original_steps=use_original_steps,
                       quantize_denoised=quantize_denoised, temperature=temperature, noise_dropout=noise_dropout,
                       score_corrector=score_corrector, corrector_kwargs=corrector_kwargs,
                       unconditional_guidance_scale=unconditional_guidance_scale,
                       noise_fn=(lambda shape: noise) if noise is not None else None)
        return st.x().clone(), pred.clone()

    @torch.no_grad()
    def stochastic_encode(self, x0, t, use_original_steps=False, noise=None):
        """sqrt(a[t]) * x0 + sqrt(1-a[t]) * noise with the ddim tables (:5378-5391); t indexes them."""
        if use_original_steps:
            sqrt_a, sqrt_1m = self.sqrt_alphas_cumprod, self.sqrt_one_minus_alphas_cumprod
        else:
            sqrt_a, sqrt_1m = torch.sqrt(self.ddim_alphas), self.ddim_sqrt_one_minus_alphas
        dev = self.model.device
        x0 = x0.to(dev, torch.float32).contiguous()
        if noise is None:
            noise = torch.randn_like(x0)
        noise = noise.to(dev, torch.float32).contiguous()
        return ops.noise_to_timesteps(x0, noise, t, sqrt_a, sqrt_1m)

    @torch.no_grad()
    def decode(self, x_latent, cond, t_start, unconditional_guidance_scale=1.0, unconditional_conditioning=None,
               use_original_steps=False, verbose=True):
        """img2img denoising loop (:5394-5412)."""
        timesteps = np.arange(self.ddpm_num_timesteps) if use_original_steps else self.ddim_timesteps
        timesteps = timesteps[:t_start]
        time_range = np.flip(timesteps)
        total_steps = timesteps.shape[0]
        if verbose:
            print(f"Running DDIM Sampling with {total_steps} timesteps")
        doubled = bool(unconditional_conditioning is not None and unconditional_conditioning is not False) \
            and unconditional_guidance_scale != 1.
        st = self._state_for(x_latent, doubled)
        for i, step in enumerate(tqdm(time_range, desc='Decoding image', total=total_steps, disable=not verbose)):
            index = total_steps - i - 1
            st.step(cond, int(step), index, use_original_steps=use_original_steps,
                    unconditional_guidance_scale=unconditional_guidance_scale, want_pred_x0=False)
        return st.x().clone()
