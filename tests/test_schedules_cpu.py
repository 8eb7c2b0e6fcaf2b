"""CPU: the PRODUCT's schedule code (LatentDiffusion.register_schedule, DDIMSampler.make_schedule) against the
tables the reference computed (G4).  These are host-side float64 -> fp32 computations and need neither the
library nor a GPU; the bar is bit equality."""
import importlib

import numpy as np
import pytest
import torch

from _util import golden

ddim = importlib.import_module("stable-diffusion_amd.ddim")
ddpm = importlib.import_module("stable-diffusion_amd.ddpm")


class _ScheduleOnly:
    """Carries the schedule buffers of a LatentDiffusion without building the device models."""
    v_posterior = 0.
    device = torch.device("cpu")


def _model():
    m = _ScheduleOnly()
    ddpm.LatentDiffusion.register_schedule(m, linear_start=0.00085, linear_end=0.012)      # txt2img.py:15-16
    return m


def test_register_schedule_buffers_equal_reference():
    m, g = _model(), golden("g4_schedules")
    assert m.num_timesteps == 1000
    for k in ("betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod",
              "sqrt_one_minus_alphas_cumprod"):
        v = getattr(m, k)
        assert v.dtype == torch.float32 and torch.equal(v, g[k]), k
    # the ancestral sampler's tables: pinned by the reference's p_sample fixture
    g11 = golden("g11_ddpm_kat")
    for k in ("posterior_variance", "posterior_log_variance_clipped", "posterior_mean_coef1", "posterior_mean_coef2",
              "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod"):
        assert torch.equal(getattr(m, k), g11["tab." + k]), k


@pytest.mark.parametrize("S", [10, 50])
@pytest.mark.parametrize("eta", [0.0, 0.5])
def test_make_schedule_tables_equal_reference(S, eta):
    g = golden("g4_schedules")
    s = ddim.DDIMSampler(_model())
    s.make_schedule(S, ddim_eta=eta, verbose=False)
    tag = f"S{S}_eta{eta}"
    assert np.array_equal(np.asarray(s.ddim_timesteps), g[tag + ".timesteps"].numpy())
    for n in ("ddim_alphas", "ddim_alphas_prev", "ddim_sigmas", "ddim_sqrt_one_minus_alphas"):
        assert np.array_equal(np.asarray(getattr(s, n), dtype=np.float64), g[tag + "." + n].numpy()), (tag, n)
    # dtypes as in the reference: fp32 tensor, float64 ndarray, float64 tensor
    assert s.ddim_alphas.dtype == torch.float32 and s.ddim_alphas_prev.dtype == np.float64
    assert s.ddim_sigmas.dtype == torch.float64
    # the scalars the fused update kernel receives are exactly the fp32 values torch.full would materialise
    a_t, a_prev, sigma, sqrt_1m = s._scalars(S - 1, False)
    assert a_t == float(np.float32(g[tag + ".ddim_alphas"][S - 1]))
    assert sigma == float(np.float32(g[tag + ".ddim_sigmas"][S - 1]))
    assert s.ddim_sigmas_for_original_num_steps.shape == (1000,)


def test_quad_discretisation_and_unknown_method():
    ts = ddim.make_ddim_timesteps("quad", 10, 1000, verbose=False)
    # int(linspace(0, sqrt(800), 10) ** 2) + 1   (stable_diffusion.py:5145-5150)
    assert ts.tolist() == [1, 10, 40, 89, 159, 247, 356, 484, 633, 801]
    with pytest.raises(NotImplementedError):
        ddim.make_ddim_timesteps("cosine", 10, 1000, verbose=False)
