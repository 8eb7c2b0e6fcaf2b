"""CPU: pin the oracle's full UNet / sampler loop / VAE against the reference's golden outputs.

The two fast cases run by default; the long ones (C5 96x96, the 37-step img2img loop, the 50-step C3
trajectory: 0.5-6 minutes each on 8 cores) run when SDMI_SLOW=1 and were run when the fixtures were made.
"""
import os

import pytest
import torch

from _util import golden, max_rel_err, psnr, structure as S, synth
from oracle import unet_ref, sampler_ref, vae_ref

slow = pytest.mark.skipif(os.environ.get("SDMI_SLOW", "0") != "1", reason="set SDMI_SLOW=1 to run")
_cache = {}


def unet_sd(cin=4):
    if cin not in _cache:
        st = S.unet_structure(**S.sd_v1_unet_config(cin))
        _cache[cin] = (st, synth.synth_state_dict(S.unet_param_shapes(st), 0))
    return _cache[cin]


def vae_sd():
    if "vae" not in _cache:
        st = S.vae_decoder_structure(S.sd_v1_ddconfig(), 4)
        _cache["vae"] = (st, synth.synth_state_dict(S.vae_decoder_param_shapes(st), 0))
    return _cache["vae"]


def test_g3_unet_c1():
    g = golden("g3_unet_c1")
    st, sd = unet_sd()
    out = unet_ref.unet_forward(sd, st, g["x"], g["t"], g["ctx"].float())
    assert max_rel_err(out, g["out"]) < 1e-5


def test_g3_unet_c2_with_taps():
    g = golden("g3_unet_c2")
    st, sd = unet_sd()
    taps = []
    out = unet_ref.unet_forward(sd, st, g["x"], g["t"], g["ctx"].float(), taps=taps)
    assert max_rel_err(out, g["out"]) < 1e-5
    assert len(taps) == 25 == g["taps"].shape[0]
    for i, t in enumerate(taps):
        ref_mean, ref_absmax, ref_l2 = (float(v) for v in g["taps"][i])
        assert abs(float(t.abs().max()) - ref_absmax) <= 1e-4 * ref_absmax, i
        assert abs(float(t.norm()) - ref_l2) <= 1e-4 * ref_l2, i


def test_g6_c1_end_to_end():
    """10-step CFG loop at 32x32 + VAE decode + clamp, all through the oracle."""
    g = golden("g6_c1_e2e")
    st, sd = unet_sd()
    vst, vsd = vae_sd()
    ctx = torch.cat([g["uc"].float(), g["c"].float()])
    eps = lambda x_in, t_in: unet_ref.apply_model(sd, st, "crossattn", x_in, t_in, ctx)
    z = sampler_ref.ddim_loop(eps, g["x_T"], 10, 0.0, 7.5, sampler_ref.ddpm_tables())
    assert max_rel_err(z, g["z"]) < 1e-4
    img = vae_ref.to_image(vae_ref.decode_first_stage(vsd, vst, z))
    assert psnr(img, g["image"].float()) > 60.0


def test_g9_inpaint_hybrid():
    g = golden("g9_inpaint")
    st, sd = unet_sd(9)
    ctx = torch.cat([g["uc"].float(), g["c"].float()])
    cond = {"c_concat": [[g["mask"]], [g["masked"]]], "c_crossattn": [ctx]}
    eps = lambda x_in, t_in: unet_ref.apply_model(sd, st, "hybrid", x_in, t_in, cond)
    z = sampler_ref.ddim_loop(eps, g["x_T"], 5, 0.0, 7.5, sampler_ref.ddpm_tables())
    assert max_rel_err(z, g["z"]) < 1e-4


@slow
def test_g3_unet_c5_hybrid_96():
    g = golden("g3_unet_c5")
    st, sd = unet_sd(9)
    out = unet_ref.diffusion_wrapper(sd, st, "hybrid", g["x"], g["t"], c_concat=[[g["mask"]], [g["masked"]]],
                                     c_crossattn=[g["ctx"].float()])
    assert max_rel_err(out, g["out"]) < 1e-5


@slow
def test_g9_img2img_decode_loop():
    g = golden("g9_img2img")
    st, sd = unet_sd()
    tab = sampler_ref.ddpm_tables()
    d = sampler_ref.ddim_tables(tab["alphas_cumprod"], 50, 0.0)
    z_enc = sampler_ref.stochastic_encode(g["init"], [37], d, g["noise"])
    assert torch.equal(z_enc, g["z_enc"])
    ctx = torch.cat([g["uc"].float(), g["c"].float()])
    eps = lambda x_in, t_in: unet_ref.apply_model(sd, st, "crossattn", x_in, t_in, ctx)
    z = sampler_ref.ddim_loop(eps, z_enc, 50, 0.0, 5.0, tab, t_start=37)
    assert max_rel_err(z, g["z"]) < 1e-4


@slow
def test_g7_c3_trajectory():
    g = golden("g7_c3_traj")
    st, sd = unet_sd()
    vst, vsd = vae_sd()
    ctx = torch.cat([g["uc"].float(), g["c"].float()])
    eps = lambda x_in, t_in: unet_ref.apply_model(sd, st, "crossattn", x_in, t_in, ctx)
    z = sampler_ref.ddim_loop(eps, g["x_T"], 50, 0.0, 7.5, sampler_ref.ddpm_tables())
    assert max_rel_err(z, g["z"]) < 1e-4
    img = vae_ref.to_image(vae_ref.decode_first_stage(vsd, vst, z))
    assert psnr(img, g["image"].float()) > 60.0
