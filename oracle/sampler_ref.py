"""fp32/float64 CPU restatement of the reference noise schedule and DDIM sampler
(TEST INFRASTRUCTURE ONLY).  File cited = /root/reference/stable_diffusion.py.
"""
import numpy as np
import torch


def beta_schedule(linear_start, linear_end, n=1000):
    """make_beta_schedule :3565-3567, "linear" = direct linspace of betas in float64."""
    return torch.linspace(linear_start, linear_end, n, dtype=torch.float64).numpy()


def ddpm_tables(linear_start=0.00085, linear_end=0.012, n=1000):
    """DDPM.register_schedule :3755-3793 (the buffers the sampler reads), float64 -> fp32."""
    betas = beta_schedule(linear_start, linear_end, n)
    alphas = 1. - betas
    ac = np.cumprod(alphas, axis=0)
    ac_prev = np.append(1., ac[:-1])
    f32 = lambda a: torch.tensor(a, dtype=torch.float32)
    return {
        "betas": f32(betas),
        "alphas_cumprod": f32(ac),
        "alphas_cumprod_prev": f32(ac_prev),
        "sqrt_alphas_cumprod": f32(np.sqrt(ac)),
        "sqrt_one_minus_alphas_cumprod": f32(np.sqrt(1. - ac)),
        "sqrt_recip_alphas_cumprod": f32(np.sqrt(1. / ac)),
        "sqrt_recipm1_alphas_cumprod": f32(np.sqrt(1. / ac - 1)),
        # v_posterior = 0 (:3782-3793)
        "posterior_variance": f32(betas * (1. - ac_prev) / (1. - ac)),
        "posterior_log_variance_clipped": f32(np.log(np.maximum(betas * (1. - ac_prev) / (1. - ac), 1e-20))),
        "posterior_mean_coef1": f32(betas * np.sqrt(ac_prev) / (1. - ac)),
        # the reference multiplies by (1 - alphas_cumprod) here (:3792-3793)
        "posterior_mean_coef2": f32(np.sqrt(alphas) * (1. - ac_prev) * (1. - ac)),
    }


def ddim_timesteps(S, n=1000):
    """make_ddim_timesteps :5141-5153, "uniform"."""
    c = n // S
    return np.asarray(list(range(0, n, c))) + 1


def ddim_tables(alphas_cumprod: torch.Tensor, S, eta):
    """make_ddim_sampling_parameters :5155-5165 + make_schedule :5205-5211.

    ``alphas_cumprod`` is the fp32 buffer; the selected values stay fp32, alphas_prev passes
    through python floats (exactly the fp32 values) and the sigma formula runs in float64 on them.
    """
    ts = ddim_timesteps(S, alphas_cumprod.shape[0])
    a = alphas_cumprod.numpy()[ts].astype(np.float32)
    a_prev = np.asarray([float(alphas_cumprod[0])] + [float(v) for v in a[:-1]])
    # :5160 mixes a float64 ndarray with an fp32 tensor: ndarray / tensor dispatches to the tensor's
    # reflected division, which torch evaluates as reciprocal(tensor) * ndarray -- i.e. 1/(1-alphas)
    # is rounded to fp32 before the float64 product.
    recip = (np.float32(1.0) / (np.float32(1.0) - a)).astype(np.float64)
    sig = eta * np.sqrt(recip * (1 - a_prev) * (1 - a.astype(np.float64) / a_prev))
    return {"timesteps": ts, "alphas": a, "alphas_prev": a_prev, "sigmas": sig,
            "sqrt_one_minus_alphas": np.sqrt(1. - a)}


def ddim_update(x, e_t, a_t, a_prev, sigma_t, sqrt_1m_at, noise, temperature=1.):
    """p_sample_ddim :5360-5375 with the four scalars already selected.

    The reference materialises each scalar with torch.full (fp32) and does all arithmetic in fp32."""
    f = lambda v: torch.full((x.shape[0], 1, 1, 1), float(v))
    a_t, a_prev, sigma_t, sqrt_1m_at = f(a_t), f(a_prev), f(sigma_t), f(sqrt_1m_at)
    pred_x0 = (x - sqrt_1m_at * e_t) / a_t.sqrt()
    dir_xt = (1. - a_prev - sigma_t ** 2).sqrt() * e_t
    n = sigma_t * noise * temperature
    x_prev = a_prev.sqrt() * pred_x0 + dir_xt + n
    return x_prev, pred_x0


def cfg_combine(e_uncond, e_cond, scale):
    """:5347."""
    return e_uncond + scale * (e_cond - e_uncond)


def stochastic_encode(x0, t_index, tab, noise):
    """DDIMSampler.stochastic_encode :5378-5391 with ddim tables (use_original_steps=False)."""
    sa = torch.sqrt(torch.as_tensor(tab["alphas"]))
    s1 = torch.as_tensor(tab["sqrt_one_minus_alphas"])
    t = torch.as_tensor(t_index, dtype=torch.long)
    shp = (x0.shape[0], 1, 1, 1)
    return sa.gather(-1, t).reshape(shp) * x0 + s1.gather(-1, t).reshape(shp) * noise


def q_sample(x_start, t, tab, noise):
    """DDPM.q_sample :3892-3895 with the 1000-step tables; ``t`` is a (b,) long tensor of timesteps."""
    shp = (x_start.shape[0],) + (1,) * (x_start.dim() - 1)
    return (tab["sqrt_alphas_cumprod"][t].reshape(shp) * x_start +
            tab["sqrt_one_minus_alphas_cumprod"][t].reshape(shp) * noise)


def known_region_blend(x0, t, tab, noise, mask, img):
    """ddim_sampling :5316-5319: the known latent noised to the step's timestep replaces the masked region."""
    return q_sample(x0, t, tab, noise) * mask + (1. - mask) * img


@torch.no_grad()
def ddim_loop(eps_model, x_T, S, eta, scale, tables_ddpm, noises=None, t_start=None, mask=None, x0=None,
              q_noises=None):
    """ddim_sampling :5276-5335 / decode :5394-5412 with the CFG contract: ``eps_model(x_in, t_in)``
    receives the doubled batch [x; x] and returns eps for [uncond; cond] (:5344-5346).

    t_start=None runs all S steps (sample); otherwise the first t_start timesteps (decode).  ``mask`` / ``x0`` /
    ``q_noises`` (one noise tensor per step) add the known-region blend of :5316-5319."""
    tab = ddim_tables(tables_ddpm["alphas_cumprod"], S, eta)
    ts = tab["timesteps"] if t_start is None else tab["timesteps"][:t_start]
    total = ts.shape[0]
    img = x_T
    b = x_T.shape[0]
    for i, step in enumerate(np.flip(ts)):
        index = total - i - 1
        t = torch.full((b,), int(step), dtype=torch.long)
        if mask is not None:
            img = known_region_blend(x0, t, tables_ddpm, q_noises[i], mask, img)
        if scale != 1.:
            e_u, e_c = eps_model(torch.cat([img] * 2), torch.cat([t] * 2)).chunk(2)
            e_t = cfg_combine(e_u, e_c, scale)
        else:
            e_t = eps_model(img, t)
        noise = noises[i] if noises is not None else torch.zeros_like(img)
        img, _ = ddim_update(img, e_t, tab["alphas"][index], tab["alphas_prev"][index],
                             tab["sigmas"][index], tab["sqrt_one_minus_alphas"][index], noise)
    return img


def ddpm_update(x, eps, t, tab, noise, clip_denoised=False, temperature=1.):
    """LatentDiffusion.p_mean_variance :4672-4684 (predict_start_from_noise :3821-3825, q_posterior
    :3839-3846) + p_sample :4710-4721 for the eps parameterization.  ``t`` is a (b,) long tensor;
    returns (x_prev, x0)."""
    shp = (x.shape[0],) + (1,) * (x.dim() - 1)
    ext = lambda name: tab[name][t].reshape(shp)
    x_recon = ext("sqrt_recip_alphas_cumprod") * x - ext("sqrt_recipm1_alphas_cumprod") * eps
    if clip_denoised:
        x_recon = x_recon.clamp(-1., 1.)
    model_mean = ext("posterior_mean_coef1") * x_recon + ext("posterior_mean_coef2") * x
    model_log_variance = ext("posterior_log_variance_clipped")
    noise = noise * temperature
    nonzero_mask = (1 - (t == 0).float()).reshape(shp)
    return model_mean + nonzero_mask * (0.5 * model_log_variance).exp() * noise, x_recon
