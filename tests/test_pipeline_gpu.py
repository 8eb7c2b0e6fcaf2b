"""GPU: the caller-visible pipeline (DDIMSampler.sample / decode / stochastic_encode, decode_first_stage,
image post-scale) through the C ABI against the reference's golden trajectories.

Bars: fp32 mode latent max-rel-err <= 1e-4 per UNet call (the loops below chain 5-10 calls, bar 5e-4);
fp16 mode final decoded image PSNR >= 35 dB after 50 DDIM steps (G7).
"""
import importlib

import pytest
import torch

from _util import golden, max_rel_err, psnr, device_latent_diffusion, drop_models

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, scope="module")
def _cleanup():
    yield
    drop_models()


def sampler_for(model):
    return importlib.import_module("stable-diffusion_amd").DDIMSampler(model)


@pytest.mark.parametrize("dt", ["fp32", "fp16"])
def test_g8_vae_decode(dt):
    model = device_latent_diffusion(4, "crossattn", dt)
    g = golden("g8_vae")
    raw = model.first_stage_model.decode(g["z"].cuda())
    dec = model.decode_first_stage(g["z"].cuda())
    torch.cuda.synchronize()
    assert tuple(dec.shape) == (1, 3, 256, 256)
    e1, e2 = max_rel_err(raw.cpu(), g["dec_raw"]), max_rel_err(dec.cpu(), g["dec"])
    print(f"vae {dt}: raw {e1:.3e} scaled {e2:.3e}")
    bar = 1e-4 if dt == "fp32" else 2e-2
    assert e1 < bar and e2 < bar


@pytest.mark.parametrize("dt", ["fp32", "fp16", "bf16"])
def test_g10_vae_encode(dt):
    """AutoencoderKL.encode -> DiagonalGaussianDistribution -> get_first_stage_encoding (img2img/inpaint entry)."""
    model = device_latent_diffusion(4, "crossattn", dt)
    g = golden("g10_vae_encode")
    post = model.encode_first_stage(g["x"].cuda())
    torch.cuda.synchronize()
    assert tuple(post.parameters.shape) == (1, 8, 16, 16)
    # bf16 keeps 8 bits of mantissa through 25 conv layers: 2e-2 is the per-UNet-call bar, the encoder lands at it
    bar = {"fp32": 1e-4, "fp16": 2e-2, "bf16": 4e-2}[dt]
    e_mom = max_rel_err(post.parameters.cpu(), g["moments"])
    z = model.get_first_stage_encoding(post, noise=g["noise"].cuda())
    z_mode = post.mode(_scale=model.scale_factor)
    e_z, e_m = max_rel_err(z.cpu(), g["z"]), max_rel_err(z_mode.cpu(), g["z_mode"])
    print(f"vae encode {dt}: moments {e_mom:.3e} z {e_z:.3e} mode {e_m:.3e}")
    assert e_mom < bar and e_z < bar and e_m < bar
    assert max_rel_err(post.mean.cpu(), g["mean"]) < bar and max_rel_err(post.std.cpu(), g["std"]) < bar
    # the fused sample kernel on the reference's own moments: only exp() may differ (by an ulp)
    imp = importlib.import_module("stable-diffusion_amd.autoencoder")
    ref_post = imp.DiagonalGaussianDistribution(g["moments"].cuda())
    zz = ref_post.sample(noise=g["noise"].cuda(), _scale=model.scale_factor).cpu()
    assert max_rel_err(zz, g["z"]) < 1e-6
    assert torch.equal(ref_post.mode().cpu(), g["mean"])
    # a second, non-square size against nothing but shape and finiteness (non-square, batch 2)
    x2 = torch.rand(2, 3, 64, 128, device="cuda") * 2 - 1     # mid attention needs h*w/64 tokens in multiples of 64
    p2 = model.encode_first_stage(x2)
    assert tuple(p2.parameters.shape) == (2, 8, 8, 16) and bool(torch.isfinite(p2.parameters).all())


@pytest.mark.parametrize("dt", ["fp32", "fp16"])
def test_g6_c1_txt2img_end_to_end(dt):
    """C1: 256x256, 10 DDIM steps, batch 1, CFG 7.5, eta 0, + VAE decode + clamp."""
    model = device_latent_diffusion(4, "crossattn", dt)
    ops = importlib.import_module("stable-diffusion_amd.ops")
    g = golden("g6_c1_e2e")
    sampler = sampler_for(model)
    cond = torch.cat([g["uc"].float(), g["c"].float()]).cuda()
    z, inter = sampler.sample(S=10, batch_size=1, shape=[4, 32, 32], conditioning=cond,
                              unconditional_guidance_scale=7.5, unconditional_conditioning=True, eta=0.,
                              x_T=g["x_T"].cuda(), verbose=False)
    img = ops.to_image(model.decode_first_stage(z))
    torch.cuda.synchronize()
    assert set(inter.keys()) == {"x_inter", "pred_x0"}
    err = max_rel_err(z.cpu(), g["z"])
    p = psnr(img.cpu(), g["image"].float())
    print(f"g6 {dt}: latent err {err:.3e}, image PSNR {p:.1f} dB")
    if dt == "fp32":
        assert err < 5e-4
        assert p > 60.0
    else:
        assert err < 5e-2
        assert p > 35.0


def test_g7_c3_50_step_trajectory_psnr_fp16():
    """512x512, 50 DDIM steps, CFG 7.5: fp16 path vs the reference fp32 image, PSNR >= 35 dB."""
    model = device_latent_diffusion(4, "crossattn", "fp16")
    ops = importlib.import_module("stable-diffusion_amd.ops")
    g = golden("g7_c3_traj")
    sampler = sampler_for(model)
    cond = torch.cat([g["uc"].float(), g["c"].float()]).cuda()
    seen = {}
    z, _ = sampler.sample(S=50, batch_size=1, shape=[4, 64, 64], conditioning=cond,
                          unconditional_guidance_scale=7.5, unconditional_conditioning=True, eta=0.,
                          x_T=g["x_T"].cuda(), verbose=False,
                          img_callback=lambda p0, i: seen.__setitem__(i, p0.clone()) if i in (0, 9, 24, 49) else None)
    img = ops.to_image(model.decode_first_stage(z))
    torch.cuda.synchronize()
    for i in (0, 9, 24, 49):
        e = max_rel_err(seen[i].cpu(), g[f"pred_x0_step{i}"])
        print(f"g7 pred_x0 step {i}: {e:.3e}")
        assert e < 5e-2
    err = max_rel_err(z.cpu(), g["z"])
    p = psnr(img.cpu(), g["image"].float())
    print(f"g7 fp16: final latent err {err:.3e}, image PSNR {p:.1f} dB")
    assert p >= 35.0


def test_c3_batch8_50_steps_graph_replay_against_golden():
    """The exact sequence bench.py times (txt2img.py:278-308 at batch 8: UNet batch 16 with the sampler's
    uniform-timestep and duplicated-pair promises, 50 DDIM steps as a replayed launch graph, CFG 7.5, VAE decode,
    clamp) held against the reference: slots 0 and 5 of the batch carry the x_T and conditioning of the reference's
    own 50-step trajectory, the other six are unrelated samples.  Image PSNR >= 35 dB against the reference's image,
    and the two slots must agree bit for bit (no cross-sample term, no slot-dependent arithmetic)."""
    pipeline = importlib.import_module("stable-diffusion_amd.pipeline")
    model = device_latent_diffusion(4, "crossattn", "fp16")
    unet = model.model.diffusion_model
    g = golden("g7_c3_traj")
    sampler = sampler_for(model)
    x_T = pipeline.initial_latents(8, 64, 64).clone()
    uc, c = pipeline.synthetic_conditioning(8)
    uc, c = uc.clone(), c.clone()
    for slot in (0, 5):
        x_T[slot], uc[slot], c[slot] = g["x_T"][0].cuda(), g["uc"][0].float().cuda(), g["c"][0].float().cuda()
    img, z = pipeline.txt2img(model, x_T, uc, c, steps=50, scale=7.5, sampler=sampler)
    torch.cuda.synchronize()
    assert unet.graph_launches() > 0, "the loop did not run as a replayed graph"
    assert tuple(img.shape) == (8, 3, 512, 512)
    p = psnr(img[0].cpu(), g["image"].float()[0])
    err = max_rel_err(z[0].cpu(), g["z"][0])
    print(f"C3 batch 8, 50 steps, graph replay ({unet.graph_launches()} launches per forward): slot 0 latent err "
          f"{err:.3e}, image PSNR {p:.1f} dB")
    assert p >= 35.0
    assert torch.equal(z[0], z[5]) and torch.equal(img[0], img[5])
    assert not torch.equal(z[0], z[1])


def test_full_size_vae_decode_batch8_against_golden():
    """The benchmark's decode shape (8 latents of 64x64 -> 8 images of 512x512): the reference's final latent of
    the 50-step trajectory, eight times; every decoded copy must match the reference's image (stored as fp16) and
    the copies must agree with each other."""
    model = device_latent_diffusion(4, "crossattn", "fp16")
    ops = importlib.import_module("stable-diffusion_amd.ops")
    g = golden("g7_c3_traj")
    img = ops.to_image(model.decode_first_stage(g["z"].repeat(8, 1, 1, 1).cuda())).cpu()
    assert tuple(img.shape) == (8, 3, 512, 512)
    ref = g["image"].float()[0]
    for i in range(8):
        assert psnr(img[i], ref) >= 50.0, (i, psnr(img[i], ref))
        assert max_rel_err(img[i], img[0]) < 1e-6
    print(f"vae decode batch 8: PSNR vs reference image {psnr(img[0], ref):.1f} dB")
    u8 = ops.to_image_u8(model.decode_first_stage(g["z"].repeat(2, 1, 1, 1).cuda())).cpu()
    assert tuple(u8.shape) == (2, 512, 512, 3) and u8.dtype == torch.uint8
    ref_u8 = (255.0 * ref.permute(1, 2, 0)).to(torch.uint8)
    assert (u8[0].int() - ref_u8.int()).abs().max() <= 2        # reference image passed through fp16 storage


@pytest.mark.parametrize("dt", ["fp32", "fp16"])
def test_img2img_pipeline_against_oracle(dt):
    """pipeline.img2img (img2img.py:269-301: encode -> posterior sample -> stochastic_encode -> decode() with
    CFG -> decode_first_stage -> clamp) against the CPU oracle composing the same calls, 128x128, 4 steps."""
    from oracle import unet_ref, sampler_ref, vae_ref
    from _util import structure as S, synth
    pipeline = importlib.import_module("stable-diffusion_amd.pipeline")
    model = device_latent_diffusion(4, "crossattn", dt)
    x = synth.uniform(51, "i2i.image", (1, 3, 128, 128), -1.0, 1.0)
    pn = synth.normal(52, "i2i.posterior_noise", (1, 4, 16, 16))
    noise = synth.normal(53, "i2i.noise", (1, 4, 16, 16))
    uc = synth.normal(2, "uncond_ctx", (1, 77, 768)).half().float()
    c = synth.normal(1, "cond_ctx.0", (1, 77, 768)).half().float()
    steps, strength, scale = 4, 0.75, 5.0
    img, z = pipeline.img2img(model, x.cuda(), uc.cuda(), c.cuda(), strength=strength, steps=steps, scale=scale,
                              noise=noise.cuda(), posterior_noise=pn.cuda())
    torch.cuda.synchronize()
    # oracle
    torch.set_num_threads(16)
    est = S.vae_encoder_structure(S.sd_v1_ddconfig(), 4)
    dst = S.vae_decoder_structure(S.sd_v1_ddconfig(), 4)
    vsd = synth.synth_state_dict(S.vae_encoder_param_shapes(est), 0)
    vsd.update(synth.synth_state_dict(S.vae_decoder_param_shapes(dst), 0))
    ust = S.unet_structure(**S.sd_v1_unet_config(4))
    usd = synth.synth_state_dict(S.unet_param_shapes(ust), 0)
    z0 = vae_ref.get_first_stage_encoding(vae_ref.encode(vsd, est, x), pn)
    ddpm = sampler_ref.ddpm_tables()
    tab = sampler_ref.ddim_tables(ddpm["alphas_cumprod"], steps, 0.0)
    t_enc = int(strength * steps)
    z_enc = sampler_ref.stochastic_encode(z0, torch.tensor([t_enc]), tab, noise)
    ctx = torch.cat([uc, c])
    z_ref = sampler_ref.ddim_loop(lambda xi, ti: unet_ref.unet_forward(usd, ust, xi, ti, ctx), z_enc, steps, 0.0, scale,
                                  ddpm, t_start=t_enc)
    img_ref = vae_ref.to_image(vae_ref.decode_first_stage(vsd, dst, z_ref))
    e_z, p = max_rel_err(z.cpu(), z_ref), psnr(img.cpu(), img_ref)
    print(f"img2img pipeline {dt}: latent {e_z:.3e}, image PSNR {p:.1f} dB")
    assert tuple(img.shape) == (1, 3, 128, 128)
    assert e_z < (5e-4 if dt == "fp32" else 5e-2) and p > (60 if dt == "fp32" else 35)


@pytest.mark.parametrize("dt", ["fp32", "fp16"])
def test_inpaint_pipeline_against_oracle(dt):
    """pipeline.inpaint (inpaint_with_text.py:51-70 make_batch, :251-291: encode image and masked image ->
    stochastic_encode at t = S-1 -> hybrid 9-channel sample() with CFG -> decode -> composite) against the CPU
    oracle composing the same calls, 128x128, 4 steps.  The masked image is formed in [0,1] and then mapped to
    [-1,1] as make_batch does, so repainted pixels enter the encoder as -1."""
    from oracle import unet_ref, sampler_ref, vae_ref
    from _util import structure as S, synth
    pipeline = importlib.import_module("stable-diffusion_amd.pipeline")
    model = device_latent_diffusion(9, "hybrid", dt)
    image01 = synth.uniform(61, "inp.image", (1, 3, 128, 128), 0.0, 1.0)
    mask = torch.zeros(1, 1, 128, 128)
    mask[:, :, 24:88, 40:120] = 1.0
    pn = (synth.normal(62, "inp.pn0", (1, 4, 16, 16)), synth.normal(63, "inp.pn1", (1, 4, 16, 16)))
    noise = synth.normal(64, "inp.noise", (1, 4, 16, 16))
    uc = synth.normal(2, "uncond_ctx", (1, 77, 768)).half().float()
    c = synth.normal(1, "cond_ctx.0", (1, 77, 768)).half().float()
    steps, scale = 4, 7.5
    image = image01 * 2.0 - 1.0
    out, z = pipeline.inpaint(model, image.cuda(), mask.cuda(), uc.cuda(), c.cuda(), steps=steps, scale=scale,
                              noise=noise.cuda(), posterior_noise=(pn[0].cuda(), pn[1].cuda()))
    torch.cuda.synchronize()
    # oracle: make_batch
    masked = ((1 - mask) * image01) * 2.0 - 1.0
    assert float(masked[:, :, 30, 50].max()) == -1.0
    torch.set_num_threads(16)
    est = S.vae_encoder_structure(S.sd_v1_ddconfig(), 4)
    dst = S.vae_decoder_structure(S.sd_v1_ddconfig(), 4)
    vsd = synth.synth_state_dict(S.vae_encoder_param_shapes(est), 0)
    vsd.update(synth.synth_state_dict(S.vae_decoder_param_shapes(dst), 0))
    ust = S.unet_structure(**S.sd_v1_unet_config(9))
    usd = synth.synth_state_dict(S.unet_param_shapes(ust), 0)
    img_latent = vae_ref.get_first_stage_encoding(vae_ref.encode(vsd, est, image), pn[0])
    masked_latent = vae_ref.get_first_stage_encoding(vae_ref.encode(vsd, est, masked), pn[1])
    img_mask = torch.nn.functional.interpolate(mask, size=img_latent.shape[-2:])
    ddpm = sampler_ref.ddpm_tables()
    tab = sampler_ref.ddim_tables(ddpm["alphas_cumprod"], steps, 0.0)
    x_T = sampler_ref.stochastic_encode(img_latent, torch.tensor([steps - 1]), tab, noise)
    cond = {"c_concat": [[torch.cat([img_mask] * 2)], [torch.cat([masked_latent] * 2)]],
            "c_crossattn": [torch.cat([uc, c])]}
    z_ref = sampler_ref.ddim_loop(lambda xi, ti: unet_ref.apply_model(usd, ust, "hybrid", xi, ti, cond), x_T, steps, 0.0,
                                  scale, ddpm)
    pred = vae_ref.to_image(vae_ref.decode_first_stage(vsd, dst, z_ref))
    out_ref = (1 - mask) * vae_ref.to_image(image) + mask * pred
    e_z, p = max_rel_err(z.cpu(), z_ref), psnr(out.cpu(), out_ref)
    print(f"inpaint pipeline {dt}: latent {e_z:.3e}, image PSNR {p:.1f} dB")
    assert tuple(out.shape) == (1, 3, 128, 128)
    assert e_z < (5e-4 if dt == "fp32" else 5e-2) and p > (60 if dt == "fp32" else 35)
    # the kept region is the input image itself
    keep = (mask == 0).expand_as(out_ref)
    assert torch.equal(out.cpu()[keep], vae_ref.to_image(image)[keep])


@pytest.mark.parametrize("dt", ["fp32", "fp16"])
def test_g9_img2img(dt):
    """stochastic_encode at strength 0.75 (t_enc = 37 of 50) then decode() with CFG 5.0."""
    model = device_latent_diffusion(4, "crossattn", dt)
    g = golden("g9_img2img")
    sampler = sampler_for(model)
    sampler.make_schedule(50, ddim_eta=0.0, verbose=False)
    z_enc = sampler.stochastic_encode(g["init"].cuda(), torch.tensor([37]), noise=g["noise"].cuda())
    assert torch.equal(z_enc.cpu(), g["z_enc"])            # elementwise fp32: bit exact
    cond = torch.cat([g["uc"].float(), g["c"].float()]).cuda()
    z = sampler.decode(z_enc, cond, 37, unconditional_guidance_scale=5.0, unconditional_conditioning=True,
                       verbose=False)
    torch.cuda.synchronize()
    err = max_rel_err(z.cpu(), g["z"])
    print(f"g9 img2img {dt}: {err:.3e}")
    assert err < (2e-3 if dt == "fp32" else 5e-2)


@pytest.mark.parametrize("dt", ["fp32", "fp16"])
def test_g9_inpaint_hybrid(dt):
    """Text-guided inpainting: hybrid conditioning dict, 9-channel UNet, 5 steps."""
    model = device_latent_diffusion(9, "hybrid", dt)
    g = golden("g9_inpaint")
    sampler = sampler_for(model)
    cond = {"c_concat": [[g["mask"].cuda()], [g["masked"].cuda()]],
            "c_crossattn": [torch.cat([g["uc"].float(), g["c"].float()]).cuda()]}
    z, _ = sampler.sample(S=5, batch_size=1, shape=[4, 32, 32], conditioning=cond,
                          unconditional_guidance_scale=7.5, unconditional_conditioning=True, eta=0.,
                          x_T=g["x_T"].cuda(), verbose=False)
    torch.cuda.synchronize()
    err = max_rel_err(z.cpu(), g["z"])
    print(f"g9 inpaint {dt}: {err:.3e}")
    assert err < (5e-4 if dt == "fp32" else 5e-2)


def test_eta_noise_path_and_p_sample_ddim_signature():
    """eta > 0 draws noise every step; with explicit noise the single-step API matches the oracle."""
    from oracle import sampler_ref
    model = device_latent_diffusion(4, "crossattn", "fp32")
    sampler = sampler_for(model)
    sampler.make_schedule(50, ddim_eta=0.5, verbose=False)
    g3 = golden("g3_unet_c1")
    x = g3["x"].cuda()
    ctx = g3["ctx"].float().cuda()
    noise = torch.randn(x.shape, generator=torch.Generator().manual_seed(5)).cuda()
    index = 45
    t = torch.full((1,), int(sampler.ddim_timesteps[index]), dtype=torch.long)
    x_prev, pred_x0 = sampler.p_sample_ddim(x, ctx, t, index, unconditional_guidance_scale=1.0, noise=noise)
    eps = model.apply_model(x, t.cuda(), ctx)
    tab = sampler_ref.ddim_tables(sampler_ref.ddpm_tables()["alphas_cumprod"], 50, 0.5)
    xp_ref, p0_ref = sampler_ref.ddim_update(x.cpu(), eps.cpu(), tab["alphas"][index], tab["alphas_prev"][index],
                                             tab["sigmas"][index], tab["sqrt_one_minus_alphas"][index], noise.cpu())
    assert torch.equal(pred_x0.cpu(), p0_ref)
    assert torch.equal(x_prev.cpu(), xp_ref)


def test_ddpm_p_sample_and_loop_against_oracle():
    """LatentDiffusion.p_sample / p_mean_variance / p_sample_loop mirrors: given the device UNet's eps, the
    ancestral update equals the oracle's bit for bit (per-sample timesteps included); a short loop runs
    with start_T and ends on the noise-free t == 0 step."""
    from oracle import sampler_ref
    model = device_latent_diffusion(4, "crossattn", "fp32")
    tab = sampler_ref.ddpm_tables()
    g3 = golden("g3_unet_c1")
    x = torch.cat([g3["x"], g3["x"].flip(-1)]).cuda()
    ctx = torch.cat([g3["ctx"].float()] * 2).cuda()
    noise = torch.randn(x.shape, generator=torch.Generator().manual_seed(9)).cuda()
    for tv, clip in (([500, 500], False), ([3, 0], True)):
        t = torch.tensor(tv, dtype=torch.long)
        eps = model.apply_model(x, t.cuda(), ctx)
        x_prev, x0 = model.p_sample(x, ctx, t, clip_denoised=clip, return_x0=True, noise=noise, temperature=0.9)
        xp_ref, x0_ref = sampler_ref.ddpm_update(x.cpu(), eps.cpu(), t, tab, noise.cpu(), clip_denoised=clip,
                                                 temperature=0.9)
        assert torch.equal(x0.cpu(), x0_ref)
        assert torch.equal(x_prev.cpu(), xp_ref)
        mean, var, logvar, x0b = model.p_mean_variance(x, ctx, t, clip_denoised=clip, return_x0=True)
        mean_ref, _ = sampler_ref.ddpm_update(x.cpu(), eps.cpu(), t, tab, torch.zeros_like(x_prev).cpu(),
                                              clip_denoised=clip)
        assert torch.equal(mean.cpu(), mean_ref) and torch.equal(x0b.cpu(), x0_ref)
        assert torch.equal(var.cpu().flatten(), tab["posterior_variance"][t])
        assert torch.equal(logvar.cpu().flatten(), tab["posterior_log_variance_clipped"][t])
    # loop: 3 steps from t = 2, deterministic last step, against the oracle driven by the same eps model
    torch.manual_seed(3)
    seen = []
    out, inter = model.p_sample_loop(ctx, tuple(x.shape), return_intermediates=True, x_T=x, start_T=3,
                                     log_every_t=1, callback=seen.append, verbose=False)
    assert seen == [2, 1, 0] and len(inter) == 4
    torch.manual_seed(3)
    img = x
    for i in (2, 1, 0):
        t = torch.full((2,), i, dtype=torch.long)
        eps = model.apply_model(img, t.cuda(), ctx)
        n = torch.randn(tuple(x.shape), device="cuda")
        img, _ = sampler_ref.ddpm_update(img.cpu(), eps.cpu(), t, tab, n.cpu())
        img = img.cuda()
    assert max_rel_err(out.cpu(), img.cpu()) < 1e-5
    # progressive_denoising: same loop, collects the x0 predictions; same RNG stream -> same result
    torch.manual_seed(3)
    out2, x0s = model.progressive_denoising(ctx, tuple(x.shape), x_T=x, start_T=3, log_every_t=1, verbose=False)
    assert torch.equal(out2, out) and len(x0s) == 3 and x0s[0].shape == x.shape


@pytest.mark.parametrize("dt", ["fp32", "fp16", "bf16"])
def test_g12_clip_text_encoder(dt):
    """FrozenCLIPEmbedder mirror on the device against the vector produced by transformers' CLIPTextModel
    (ViT-L/14 text configuration, synthesised weights); state-dict key layouts of an SD checkpoint and of
    current transformers are both accepted."""
    sdmi = importlib.import_module("stable-diffusion_amd")
    from _util import synth
    g = golden("g12_clip_text")
    enc = sdmi.FrozenCLIPEmbedder(compute_dtype=dt)
    shapes = sdmi.clip_text_param_shapes()
    assert enc.state_dict_shapes() == shapes
    sd = {"cond_stage_model.transformer." + k: synth.synth_tensor(0, k, s, "cuda") for k, s in shapes.items()}
    _, unexpected = enc.load_state_dict(sd, prefix="cond_stage_model.")
    assert not unexpected
    out = enc.encode(g["ids"])
    assert out.shape == (3, 77, 768) and out.dtype == torch.float32
    err = max_rel_err(out.cpu(), g["last_hidden_state"])
    print(f"g12 clip text {dt}: {err:.3e}")
    assert err < {"fp32": 1e-4, "fp16": 2e-2, "bf16": 4e-2}[dt]
    # a shorter sequence (no padding) uses the first positions only and is causal: equal to the prefix
    short = enc(g["ids"][:, :20].cuda())
    assert max_rel_err(short.cpu(), out[:, :20].cpu()) < {"fp32": 1e-5, "fp16": 5e-3, "bf16": 4e-2}[dt]
    with pytest.raises(IndexError):
        enc(torch.full((1, 77), 49408, dtype=torch.long))
    with pytest.raises(RuntimeError, match="tokenizer"):
        enc("a photograph of an astronaut riding a horse")
    if dt == "fp32":
        enc2 = sdmi.FrozenCLIPEmbedder(compute_dtype=dt)
        enc2.load_state_dict({k[len("text_model."):]: synth.synth_tensor(0, k, s, "cuda") for k, s in shapes.items()})
        assert torch.equal(enc2(g["ids"]), out)


def test_txt2img_from_token_ids():
    """The script-side sequence with the text transformer inside the model: token ids -> get_learned_conditioning
    -> sampler -> decode equals the same run fed with the encoder's hidden states; a model without text
    transformer weights refuses token ids."""
    pipeline = importlib.import_module("stable-diffusion_amd.pipeline")
    model = pipeline.build_synthetic_model(4, "crossattn", "fp16", text_encoder=True)
    uc_ids, c_ids = pipeline.synthetic_prompt_tokens(2)
    assert uc_ids.shape == (2, 77) and int(c_ids[0, 0]) == 49406 and int(c_ids[1, -1]) == 49407
    uc, c = model.get_learned_conditioning(uc_ids), model.get_learned_conditioning(c_ids)
    assert c.shape == (2, 77, 768) and c.dtype == torch.float32
    assert torch.equal(model.get_learned_conditioning(c), c)            # hidden states pass through
    x_T = pipeline.initial_latents(2, 16, 16)
    img_a, z_a = pipeline.txt2img(model, x_T, uc_ids, c_ids, steps=4)
    img_b, z_b = pipeline.txt2img(model, x_T, uc, c, steps=4)
    assert torch.equal(z_a, z_b) and torch.equal(img_a, img_b)
    assert bool(torch.isfinite(img_a).all()) and img_a.shape == (2, 3, 128, 128)
    del model
    plain = device_latent_diffusion(4, "crossattn", "fp16")
    with pytest.raises(NotImplementedError):
        plain.get_learned_conditioning(c_ids)


@pytest.mark.parametrize("dt", ["fp32", "fp16"])
def test_vae_sizes_whose_token_count_is_no_multiple_of_64(dt):
    """The reference's autoencoder takes any latent size (Decoder.forward :496 has no size condition, the encoder
    needs multiples of 8 only); the mid-block attention's two-GEMM path contracts over the tokens in whole K-tiles,
    other token counts (here 35, 54 and 45) take the vector-ALU attention kernel.  Against the fp32 CPU oracle."""
    from oracle import vae_ref
    from _util import structure as S, synth
    model = device_latent_diffusion(4, "crossattn", dt)
    est = S.vae_encoder_structure(S.sd_v1_ddconfig(), 4)
    dst = S.vae_decoder_structure(S.sd_v1_ddconfig(), 4)
    vsd = synth.synth_state_dict(S.vae_encoder_param_shapes(est), 0)
    vsd.update(synth.synth_state_dict(S.vae_decoder_param_shapes(dst), 0))
    torch.set_num_threads(16)
    bar = 1e-4 if dt == "fp32" else 2e-2
    for shape in ((1, 4, 5, 7), (2, 4, 9, 6)):
        z = synth.normal(61, f"oddvae.z.{shape[2]}x{shape[3]}", shape)
        dec = model.first_stage_model.decode(z.cuda())
        torch.cuda.synchronize()
        ref = vae_ref.decode(vsd, dst, z)
        assert tuple(dec.shape) == (shape[0], 3, 8 * shape[2], 8 * shape[3])
        e = max_rel_err(dec.cpu(), ref)
        print(f"vae decode {shape} {dt}: {e:.3e}")
        assert e < bar
    x = synth.uniform(62, "oddvae.image", (1, 3, 72, 40), -1.0, 1.0)
    post = model.encode_first_stage(x.cuda())
    torch.cuda.synchronize()
    e = max_rel_err(post.parameters.cpu(), vae_ref.encode(vsd, est, x))
    print(f"vae encode 72x40 {dt}: {e:.3e}")
    assert tuple(post.parameters.shape) == (1, 8, 9, 5) and e < bar
