"""GPU: known answers of the reference that exercise the PRODUCT's classes end to end through the C ABI:

* whole blocks through the executor's own code path (sdmi_unet_run_block): the reference's ResBlock and
  SpatialTransformer outputs (G2);
* LatentDiffusion.q_sample / DDIMSampler.stochastic_encode against the reference's outputs (G5), including the
  one-element timestep tensor the inpainting script passes;
* the known-region blend of DDIMSampler.sample(mask=, x0=) (G13): the kernel bit for bit, the loop within the
  per-call bar.

Bars: bit equality for the fp32 elementwise kernels; max|a-b|/max|b| <= 1e-4 per UNet call in fp32 mode (the
4-step loop: 5e-4), <= 2e-2 in fp16 / bf16 mode (block tests: the per-op tolerances of test_ops_gpu).
"""
import importlib
from collections import OrderedDict

import pytest
import torch

from _util import golden, max_rel_err, g2_u, structure as S, synth, device_latent_diffusion, drop_models

pytestmark = pytest.mark.gpu

TOL = {"fp32": 1e-4, "fp16": 6e-3, "bf16": 4e-2}


def tdt(name):
    return {"fp32": torch.float32, "fp16": torch.float16, "bf16": torch.bfloat16}[name]


def nhwc(x, dt):
    return x.permute(0, 2, 3, 1).contiguous().cuda().to(tdt(dt))


def nchw(y):
    return y.float().cpu().permute(0, 3, 1, 2).contiguous()


@pytest.fixture(autouse=True, scope="module")
def _cleanup():
    yield
    drop_models()


_nets = {}


def unet_with_g2_blocks(dt):
    """SD-v1 UNet whose first transformer (input_blocks.1.1), first ResBlock (input_blocks.1.0, 320 -> 320) and
    ResBlock output_blocks.11.0 (640 -> 320 with a 1x1 skip convolution) carry the weights the reference's
    stand-alone G2 modules were given."""
    if dt not in _nets:
        sdmi = importlib.import_module("stable-diffusion_amd")
        cfg = S.sd_v1_unet_config(4)
        net = sdmi.UNetModel(**cfg, compute_dtype=dt)
        shapes = net.state_dict_shapes()
        sd = synth.synth_state_dict(shapes, 0, device="cuda")
        for dst, src in (("input_blocks.1.1.", "g2.st."), ("input_blocks.1.0.", "g2.res_a."),
                         ("output_blocks.11.0.", "g2.res_b.")):
            for k, shp in shapes.items():
                if k.startswith(dst):
                    sd[k] = synth.synth_tensor(0, src + k[len(dst):], shp, "cuda")
        net.load_state_dict(sd)
        _nets[dt] = net
    return _nets[dt]


def _block_index(net, kind, prefix):
    tail = ".in_layers.2.weight" if kind == "res" else ".proj_out.weight"
    names = [k[:-len(tail)] for k in net.state_dict_shapes() if k.endswith(tail)]
    return names.index(prefix)


@pytest.mark.parametrize("dt", ["fp32", "fp16", "bf16"])
def test_g2_spatial_transformer_through_the_executor(dt):
    """SpatialTransformer(320, 8 heads, d_head 40, context 768).forward(x, context) (:2885-2896)."""
    g = golden("g2_blocks")
    net = unet_with_g2_blocks(dt)
    x = nhwc(g2_u("st.x", (2, 320, 8, 8)), dt)
    ctx = g2_u("ctx", (2, 77, 768)).cuda()
    y = net.run_block("st", _block_index(net, "st", "input_blocks.1.1"), x, context=ctx)
    torch.cuda.synchronize()
    err = max_rel_err(nchw(y), g["st"])
    print(f"g2 st {dt}: {err:.3e}")
    assert err < TOL[dt]


@pytest.mark.parametrize("dt", ["fp32", "fp16", "bf16"])
def test_g2_resblocks_through_the_executor(dt):
    """ResBlock.forward(x, emb) (:3197-3234): identity skip (320 -> 320) and 1x1-convolution skip (640 -> 320),
    the latter also with its input handed over as two channel-concatenated tensors like an output block's."""
    g = golden("g2_blocks")
    net = unet_with_g2_blocks(dt)
    emb = g2_u("emb", (2, 1280)).cuda()
    xa = nhwc(g2_u("res_a.x", (2, 320, 8, 8)), dt)
    ya = net.run_block("res", _block_index(net, "res", "input_blocks.1.0"), xa, emb=emb)
    xb = g2_u("res_b.x", (2, 640, 8, 8))
    ib = _block_index(net, "res", "output_blocks.11.0")
    yb = net.run_block("res", ib, nhwc(xb, dt), emb=emb)
    yb2 = net.run_block("res", ib, nhwc(xb[:, :320], dt), emb=emb, skip=nhwc(xb[:, 320:], dt))
    torch.cuda.synchronize()
    ea, eb, eb2 = max_rel_err(nchw(ya), g["res_a"]), max_rel_err(nchw(yb), g["res_b"]), max_rel_err(nchw(yb2), g["res_b"])
    print(f"g2 resblocks {dt}: {ea:.3e} {eb:.3e} {eb2:.3e}")
    assert ea < TOL[dt] and eb < TOL[dt] and eb2 < TOL[dt]


@pytest.mark.parametrize("dt", ["fp16", "bf16"])
def test_layernorm_row_sums_from_the_producing_gemm(dt, monkeypatch):
    """In the 16-bit modes the LayerNorms of a transformer block take their row statistics from sums that the GEMM
    which wrote their input leaves behind (large-tile epilogue, split-K reduction, or a fallback pass, depending on
    the shape) instead of a pass of their own: same result as with the separate statistics kernel, for a map that
    takes the large-tile kernels (64x64), one that takes split-K (8x8) and one in between."""
    sdmi = importlib.import_module("stable-diffusion_amd")
    cfg = S.sd_v1_unet_config(4)
    nets = []
    for flag in ("1", "0"):
        monkeypatch.setenv("SDMI_LN_PRODUCER_STATS", flag)
        net = sdmi.UNetModel(**cfg, compute_dtype=dt)
        net.load_state_dict(synth.synth_state_dict(net.state_dict_shapes(), 0, device="cuda"))
        nets.append(net)
    ctx = g2_u("ctx", (2, 77, 768)).cuda()
    for name, hw in (("input_blocks.1.1", 64), ("input_blocks.4.1", 20), ("middle_block.1", 8)):
        idx = _block_index(nets[0], "st", name)
        ch = {"input_blocks.1.1": 320, "input_blocks.4.1": 640, "middle_block.1": 1280}[name]
        x = nhwc(synth.normal(5, f"ps.{name}", (2, ch, hw, hw)), dt)
        ya = nets[0].run_block("st", idx, x, context=ctx)
        yb = nets[1].run_block("st", idx, x, context=ctx)
        torch.cuda.synchronize()
        err = max_rel_err(ya.float().cpu(), yb.float().cpu())
        print(f"row sums {dt} {name} {hw}x{hw}: {err:.3e}")
        assert err < TOL[dt]
    del nets


@pytest.mark.parametrize("dt", ["fp16", "bf16"])
def test_layernorm_row_sums_with_a_large_common_offset(dt, monkeypatch):
    """The row sums the producing GEMM leaves are plain (sum, sum of squares) in fp32, so the variance is a
    difference of two large numbers when a row sits far from zero.  Rows of mean ~50 and spread ~0.6 (proj_in's
    bias raised by 50: outlier channels of real checkpoints look like this) must still give the result of the
    separate statistics kernel, which shifts the data first."""
    sdmi = importlib.import_module("stable-diffusion_amd")
    cfg = S.sd_v1_unet_config(4)
    nets = []
    for flag in ("1", "0"):
        monkeypatch.setenv("SDMI_LN_PRODUCER_STATS", flag)
        net = sdmi.UNetModel(**cfg, compute_dtype=dt)
        sd = synth.synth_state_dict(net.state_dict_shapes(), 0, device="cuda")
        for name in ("input_blocks.1.1", "middle_block.1"):
            sd[name + ".proj_in.bias"] = sd[name + ".proj_in.bias"] + 50.0
        net.load_state_dict(sd)
        nets.append(net)
    ctx = g2_u("ctx", (2, 77, 768)).cuda()
    for name, hw, ch in (("input_blocks.1.1", 64, 320), ("middle_block.1", 8, 1280)):
        idx = _block_index(nets[0], "st", name)
        x = nhwc(synth.normal(5, f"ps.{name}", (2, ch, hw, hw)), dt)
        ya = nets[0].run_block("st", idx, x, context=ctx)
        yb = nets[1].run_block("st", idx, x, context=ctx)
        torch.cuda.synchronize()
        err = max_rel_err(ya.float().cpu(), yb.float().cpu())
        print(f"row sums with offset {dt} {name} {hw}x{hw}: {err:.3e}")
        assert err < TOL[dt]
    del nets


@pytest.mark.parametrize("dt", ["fp16", "bf16"])
def test_cross_attention_fold_equals_the_three_launch_path(dt, monkeypatch):
    """In the 16-bit modes the cross-attention of a transformer block runs as two GEMMs with weights folded once
    per prompt (scores = LayerNorm(x) G^T with a per-head softmax in the epilogue, then P U^T + bias + x) in place of
    to_q -> attention -> to_out.  Same function, other rounding points: the two paths must agree within the block
    tolerance on maps that take the 256-, 128- and 64-row tiles, and both must stay within it of each other on a
    map whose token count no tile divides (20x20: both nets take the three-launch path there)."""
    sdmi = importlib.import_module("stable-diffusion_amd")
    cfg = S.sd_v1_unet_config(4)
    nets = []
    monkeypatch.setenv("SDMI_XATTN_FOLD_MIN_D", "0")          # (by default the 320-wide level keeps three launches)
    for flag in ("1", "0"):
        monkeypatch.setenv("SDMI_XATTN_FOLD", flag)
        net = sdmi.UNetModel(**cfg, compute_dtype=dt)
        net.load_state_dict(synth.synth_state_dict(net.state_dict_shapes(), 0, device="cuda"))
        nets.append(net)
    ctx = g2_u("ctx", (2, 77, 768)).cuda()
    for name, hw, ch in (("input_blocks.1.1", 64, 320), ("input_blocks.4.1", 32, 640), ("input_blocks.7.1", 16, 1280),
                         ("middle_block.1", 8, 1280), ("input_blocks.4.1", 20, 640)):
        idx = _block_index(nets[0], "st", name)
        x = nhwc(synth.normal(5, f"xf.{name}", (2, ch, hw, hw)), dt)
        ya = nets[0].run_block("st", idx, x, context=ctx)
        yb = nets[1].run_block("st", idx, x, context=ctx)
        torch.cuda.synchronize()
        err = max_rel_err(ya.float().cpu(), yb.float().cpu())
        print(f"cross-attention fold {dt} {name} {hw}x{hw}: {err:.3e}")
        assert err < TOL[dt]
    del nets


@pytest.mark.parametrize("dt", ["fp16", "bf16"])
def test_skip_connection_folded_into_the_second_convolution(dt, monkeypatch):
    """ResBlocks that change the channel count return skip_connection(x) + conv(h) (stable_diffusion.py:3234); in
    the 16-bit modes the 1x1 skip convolution rides as extra K columns of the 3x3 convolution (one launch, one
    rounding of the sum instead of two).  Against the two-launch path: a block with a concatenated input at 64x64
    (tap-inner walk), one at 32x32 (tap-major), one at 8x8 (split-K) and one that only changes the width (in4)."""
    sdmi = importlib.import_module("stable-diffusion_amd")
    cfg = S.sd_v1_unet_config(4)
    nets = []
    for flag in ("1", "0"):
        monkeypatch.setenv("SDMI_FUSE_SKIP", flag)
        net = sdmi.UNetModel(**cfg, compute_dtype=dt)
        net.load_state_dict(synth.synth_state_dict(net.state_dict_shapes(), 0, device="cuda"))
        nets.append(net)
    for name, hw, ch, skip in (("output_blocks.11.0", 64, 320, 320), ("output_blocks.9.0", 64, 640, 320),
                               ("output_blocks.6.0", 32, 1280, 640), ("output_blocks.1.0", 8, 1280, 1280),
                               ("input_blocks.4.0", 32, 320, 0), ("output_blocks.5.0", 12, 1280, 640)):
        idx = _block_index(nets[0], "res", name)
        x = nhwc(synth.normal(5, f"fs.{name}", (2, ch, hw, hw)), dt)
        sk = nhwc(synth.normal(6, f"fs.skip.{name}", (2, skip, hw, hw)), dt) if skip else None
        emb = synth.normal(7, f"fs.emb.{name}", (2, 1280)).cuda()
        ya = nets[0].run_block("res", idx, x, emb=emb, skip=sk)
        yb = nets[1].run_block("res", idx, x, emb=emb, skip=sk)
        torch.cuda.synchronize()
        err = max_rel_err(ya.float().cpu(), yb.float().cpu())
        print(f"skip fold {dt} {name} {hw}x{hw}: {err:.3e}")
        assert err < TOL[dt]
    del nets


def test_q_sample_and_stochastic_encode_g5_bit_exact():
    """LatentDiffusion.q_sample (:3892-3895) with per-sample timesteps and DDIMSampler.stochastic_encode
    (:5378-5391), both with a timestep per sample and with the one-element tensor of inpaint_with_text.py:260."""
    model = device_latent_diffusion(4, "crossattn", "fp32")
    g = golden("g5_sampler_kat")
    x, noise = g["x"].cuda(), g["noise"].cuda()
    out = model.q_sample(x, torch.tensor([981, 1]), noise=noise)
    assert torch.equal(out.cpu(), g["q_sample.t981_1"])
    one = model.q_sample(x, torch.tensor([981]), noise=noise)            # a single t serves the whole batch
    assert torch.equal(one[0].cpu(), g["q_sample.t981_1"][0])
    sampler = importlib.import_module("stable-diffusion_amd").DDIMSampler(model)
    sampler.make_schedule(50, ddim_eta=0.0, verbose=False)
    for tidx in (36, 49):
        ref = g[f"stochastic_encode.t{tidx}"]
        assert torch.equal(sampler.stochastic_encode(x, torch.tensor([tidx, tidx]), noise=noise).cpu(), ref)
        assert torch.equal(sampler.stochastic_encode(x, torch.tensor([tidx]), noise=noise).cpu(), ref)
    with pytest.raises(AssertionError, match="timesteps for a batch"):
        model.q_sample(x, torch.tensor([1, 2, 3]), noise=noise)


def test_known_region_blend_kernel_g13_bit_exact():
    ops = importlib.import_module("stable-diffusion_amd.ops")
    model = device_latent_diffusion(4, "crossattn", "fp32")
    g = golden("g13_mask_blend")
    t = 751                                              # first step of S = 4: timesteps 1, 251, 501, 751
    out = ops.q_sample_blend(g["x0"].cuda(), g["q_noise"][0].cuda(), g["mask"].cuda(), g["x_T"].cuda(),
                             model.sqrt_alphas_cumprod[t], model.sqrt_one_minus_alphas_cumprod[t])
    assert torch.equal(out.cpu(), g["blended"][0])
    # a mask with one plane per channel gives the same result as the broadcast one; in place is allowed
    img = g["x_T"].cuda().clone()
    ops.q_sample_blend(g["x0"].cuda(), g["q_noise"][0].cuda(), g["mask"].repeat(1, 4, 1, 1).cuda(), img,
                       model.sqrt_alphas_cumprod[t], model.sqrt_one_minus_alphas_cumprod[t], out=img)
    assert torch.equal(img.cpu(), g["blended"][0])
    with pytest.raises(RuntimeError, match="mask has 2 channels"):
        ops.q_sample_blend(g["x0"].cuda(), g["q_noise"][0].cuda(), g["mask"].repeat(1, 2, 1, 1).cuda(),
                           g["x_T"].cuda(), 1.0, 0.0)


@pytest.mark.parametrize("dt", ["fp32", "fp16"])
def test_g13_sample_with_known_region(dt):
    """DDIMSampler.sample(mask=, x0=) (:5316-5319), 4 steps with CFG 7.5; q_sample's noise injected."""
    model = device_latent_diffusion(4, "crossattn", dt)
    g = golden("g13_mask_blend")
    sampler = importlib.import_module("stable-diffusion_amd").DDIMSampler(model)
    noises = [n.cuda() for n in g["q_noise"]]
    cond = torch.cat([g["uc"].float(), g["c"].float()]).cuda()
    z, _ = sampler.sample(S=4, batch_size=1, shape=[4, 16, 16], conditioning=cond, mask=g["mask"].cuda(),
                          x0=g["x0"].cuda(), unconditional_guidance_scale=7.5, unconditional_conditioning=True,
                          eta=0., x_T=g["x_T"].cuda(), verbose=False, q_noise_fn=lambda shape: noises.pop(0))
    torch.cuda.synchronize()
    assert not noises
    err = max_rel_err(z.cpu(), g["z"])
    print(f"g13 {dt}: {err:.3e}")
    assert err < (5e-4 if dt == "fp32" else 5e-2)


def _oracle_unet_sd():
    from oracle import unet_ref
    st = S.unet_structure(**S.sd_v1_unet_config(4))
    return unet_ref, st, synth.synth_state_dict(S.unet_param_shapes(st), 0)


def test_p_sample_ddim_per_sample_timesteps_against_oracle():
    """p_sample_ddim honours a timestep per sample (:5344-5346: t_in = cat([t] * 2) goes to the model, the update's
    scalars come from `index`): fp32 mode against the oracle's UNet + guidance combine + update, <= 1e-4; a batch
    whose timesteps are all equal takes the single-row embedding path and must agree with the per-row path."""
    from oracle import sampler_ref
    unet_ref, st, sd = _oracle_unet_sd()
    model = device_latent_diffusion(4, "crossattn", "fp32")
    sampler = importlib.import_module("stable-diffusion_amd").DDIMSampler(model)
    sampler.make_schedule(50, ddim_eta=0.0, verbose=False)
    x = synth.normal(11, "psd.x", (2, 4, 16, 16))
    ctx = torch.cat([synth.normal(2, "psd.uc", (2, 77, 768)), synth.normal(1, "psd.c", (2, 77, 768))])
    t = torch.tensor([981, 961])
    index = 49
    x_prev, pred = sampler.p_sample_ddim(x.cuda(), ctx.cuda(), t.cuda(), index, unconditional_guidance_scale=7.5,
                                         unconditional_conditioning=True)
    torch.cuda.synchronize()
    eps = unet_ref.unet_forward(sd, st, torch.cat([x] * 2), torch.cat([t] * 2), ctx)
    e_u, e_c = eps.chunk(2)
    tab = sampler_ref.ddim_tables(sampler_ref.ddpm_tables()["alphas_cumprod"], 50, 0.0)
    xp_ref, p0_ref = sampler_ref.ddim_update(x, sampler_ref.cfg_combine(e_u, e_c, 7.5), tab["alphas"][index],
                                             tab["alphas_prev"][index], tab["sigmas"][index],
                                             tab["sqrt_one_minus_alphas"][index], torch.zeros_like(x))
    ex, ep = max_rel_err(x_prev.cpu(), xp_ref), max_rel_err(pred.cpu(), p0_ref)
    print(f"p_sample_ddim per-sample t: x_prev {ex:.3e} pred_x0 {ep:.3e}")
    assert ex < 1e-4 and ep < 1e-4
    # equal timesteps: the uniform-timestep shortcut and the per-row embedding agree
    st_a = sampler._StepState(sampler, x.cuda(), True)
    st_a.step(ctx.cuda(), 981, index, unconditional_guidance_scale=7.5)
    st_b = sampler._StepState(sampler, x.cuda(), True)
    st_b.step(ctx.cuda(), torch.tensor([981, 981]), index, unconditional_guidance_scale=7.5)
    assert max_rel_err(st_a.x().cpu(), st_b.x().cpu()) < 1e-5


def test_ddim_score_corrector_and_noise_dropout():
    """The two side branches of p_sample_ddim: `score_corrector.modify_score` receives the guided score and what it
    returns drives the update (:5351-5353), checked against the oracle's update on the corrected score; and
    `noise_dropout` (:5372-5373) zeroes a share of the noise term and scales the rest by 1 / (1 - p)."""
    from oracle import sampler_ref
    ops = importlib.import_module("stable-diffusion_amd.ops")
    model = device_latent_diffusion(4, "crossattn", "fp32")
    sampler = importlib.import_module("stable-diffusion_amd").DDIMSampler(model)
    sampler.make_schedule(50, ddim_eta=0.5, verbose=False)
    x = synth.normal(11, "psd.x", (2, 4, 16, 16)).cuda()
    ctx = torch.cat([synth.normal(2, "psd.uc", (2, 77, 768)), synth.normal(1, "psd.c", (2, 77, 768))]).cuda()
    t = torch.tensor([981, 981]).cuda()
    index = 49
    seen = {}

    class Corrector:
        def modify_score(self, mdl, e_t, xx, tt, c, gain=1.0):
            seen.update(e=e_t.clone(), x=xx.clone(), t=tt.clone(), gain=gain, model=mdl)
            return gain * e_t + 0.25 * xx

    noise = synth.normal(12, "psd.noise", (2, 4, 16, 16)).cuda()
    x_prev, pred = sampler.p_sample_ddim(x, ctx, t, index, unconditional_guidance_scale=7.5,
                                         unconditional_conditioning=True, score_corrector=Corrector(),
                                         corrector_kwargs={"gain": 0.5}, noise=noise)
    torch.cuda.synchronize()
    assert seen["model"] is model and seen["gain"] == 0.5
    assert torch.equal(seen["x"], x) and torch.equal(seen["t"], t)
    # the score handed over is the guided one: the combine kernel on the UNet's output for the doubled batch
    # (the sampler's call runs with its uniform-timestep / duplicated-pair promises on, which may pick other GEMM
    # plans than this plain call: equal within the fp32 bar, while the combine kernel itself is bit-exact)
    eps2 = model.apply_model(torch.cat([x] * 2), torch.cat([t] * 2), ctx)
    e_u, e_c = eps2.cpu().chunk(2)
    assert torch.equal(ops.cfg_combine(eps2, 7.5).cpu(), sampler_ref.cfg_combine(e_u, e_c, 7.5))
    assert max_rel_err(seen["e"].cpu(), sampler_ref.cfg_combine(e_u, e_c, 7.5)) < 1e-4
    tab = sampler_ref.ddim_tables(sampler_ref.ddpm_tables()["alphas_cumprod"], 50, 0.5)
    e_mod = 0.5 * seen["e"].cpu() + 0.25 * x.cpu()
    xp_ref, p0_ref = sampler_ref.ddim_update(x.cpu(), e_mod, tab["alphas"][index], tab["alphas_prev"][index],
                                             tab["sigmas"][index], tab["sqrt_one_minus_alphas"][index], noise.cpu())
    assert torch.equal(x_prev.cpu(), xp_ref) and torch.equal(pred.cpu(), p0_ref)
    # noise dropout: with p = 0.5 every element's noise term is either absent or doubled
    st = sampler._StepState(sampler, x, True)
    st.step(ctx, 981, index, unconditional_guidance_scale=7.5, noise_fn=lambda shape: noise, noise_dropout=0.5)
    st0 = sampler._StepState(sampler, x, True)
    st0.step(ctx, 981, index, unconditional_guidance_scale=7.5, noise_fn=lambda shape: torch.zeros_like(noise))
    term = (st.x() - st0.x()).cpu()
    full = float(tab["sigmas"][index]) * noise.cpu()
    dropped = term.abs() < 1e-6
    doubled = (term - 2.0 * full).abs() < 1e-4 * full.abs().max()
    assert bool((dropped | doubled).all())
    assert 0.3 < float(dropped.float().mean()) < 0.7
    with pytest.raises(NotImplementedError, match="VQ first stage"):
        sampler.p_sample_ddim(x, ctx, t, index, quantize_denoised=True)


def test_default_compute_dtype_follows_use_fp16():
    """The reference computes in fp32 unless use_fp16 is set (:3302); so does the mirror unless told otherwise."""
    unet = importlib.import_module("stable-diffusion_amd.unet")
    cfg = S.sd_v1_unet_config(4)
    assert unet.UNetModel(**cfg).compute_dtype == "fp32"
    assert unet.UNetModel(**dict(cfg, use_fp16=True)).compute_dtype == "fp16"
    assert unet.UNetModel(**cfg, compute_dtype="bf16").compute_dtype == "bf16"
