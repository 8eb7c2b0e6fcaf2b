"""The caller-side sequence of the reference's txt2img script (txt2img.py:278-308), on the device,
plus the sharding / gathering used for multi-GPU runs.

    x_T -> DDIMSampler.sample(S, conditioning=cat([uc, c]), scale, eta, x_T) -> decode_first_stage
        -> clamp((x + 1) / 2, 0, 1)

Samples (prompt x seed pairs) are independent trajectories, so a multi-GPU run gives each rank a
contiguous block of them, runs the whole loop locally with a full model replica and exchanges
nothing until the decoded images are all-gathered (RCCL) at the end.
"""
import torch

from . import synth
from .structure import sd_v1_model_params


def build_synthetic_model(in_channels=4, conditioning_key="crossattn", dtype="fp16", seed=0, device="cuda",
                          encoder=False, text_encoder=False):
    """LatentDiffusion with SD-v1-shaped weights from the deterministic synthesiser (no checkpoint
    exists offline).  Hash keys are the un-prefixed checkpoint names.  ``encoder`` adds the VAE encoder,
    ``text_encoder`` the CLIP ViT-L/14 text transformer (``cond_stage_model.transformer.*``)."""
    from .ddpm import LatentDiffusion
    params = sd_v1_model_params(in_channels, conditioning_key)
    model = LatentDiffusion(**params, compute_dtype=dtype, device=device)
    sd = {"model.diffusion_model." + k: synth.synth_tensor(seed, k, s, device)
          for k, s in model.model.diffusion_model.state_dict_shapes().items()}
    sd.update({"first_stage_model." + k: synth.synth_tensor(seed, k, s, device)
               for k, s in model.first_stage_model.state_dict_shapes(encoder=encoder).items()})
    if text_encoder:
        from .clip import clip_text_param_shapes
        sd.update({"cond_stage_model.transformer." + k: synth.synth_tensor(seed, k, s, device)
                   for k, s in clip_text_param_shapes().items()})
    model.load_state_dict(sd)
    return model


def synthetic_prompt_tokens(batch, device="cuda", first_sample=0, tokens=77, vocab=49408):
    """Stand-ins for the tokenizer's output (its vocabulary is a download): per global sample index, BOS, a
    run of 6..37 pseudo-random word ids, then EOS padding to `tokens`; and the empty prompt (BOS, EOS...)."""
    c = torch.full((batch, tokens), vocab - 1, dtype=torch.long)
    c[:, 0] = vocab - 2
    for i in range(batch):
        g = first_sample + i
        n = 6 + g % 32
        c[i, 1:1 + n] = (synth.uniform01(3, f"prompt.{g}", n) * (vocab - 2)).long()
    uc = torch.full((batch, tokens), vocab - 1, dtype=torch.long)
    uc[:, 0] = vocab - 2
    return uc.to(device), c.to(device)


def synthetic_conditioning(batch, device="cuda", first_sample=0):
    """Stand-ins for the CLIP hidden states of `batch` prompts and of the empty prompt: N(0,1),
    rounded to fp16-representable values, one stream per global sample index."""
    c = torch.cat([synth.normal(1, f"cond_ctx.{first_sample + i}", (1, 77, 768)) for i in range(batch)])
    uc = synth.normal(2, "uncond_ctx", (1, 77, 768)).repeat(batch, 1, 1)
    return uc.half().float().to(device), c.half().float().to(device)


def initial_latents(batch, h, w, device="cuda", first_sample=0, seed=42):
    """x_T per global sample index, so results do not depend on how samples are sharded."""
    return torch.cat([synth.normal(seed, f"x_T.{first_sample + i}", (1, 4, h, w)) for i in range(batch)]).to(device)


@torch.no_grad()
def txt2img(model, x_T, uc, c, steps=50, scale=7.5, eta=0.0, sampler=None, as_uint8=False):
    """One batch through the whole path; returns images (B,3,8h,8w) in [0,1] and the final latents.
    ``uc`` / ``c`` are conditioning tensors (B,77,768), or token ids / text for the model's text transformer
    (txt2img.py:285-290: get_learned_conditioning of the empty prompt and of the prompts).  ``as_uint8``: the
    images come back as the (B,8h,8w,3) uint8 array the script hands to Image.fromarray (txt2img.py:307-317), a
    quarter of the bytes for a multi-GPU gather."""
    from . import ops
    from .ddim import DDIMSampler
    sampler = sampler or DDIMSampler(model)
    b, _, h, w = x_T.shape
    c = model.get_learned_conditioning(c)
    if scale != 1.0:
        uc = model.get_learned_conditioning(uc)
    cond = torch.cat([uc, c]) if scale != 1.0 else c
    z, _ = sampler.sample(S=steps, batch_size=b, shape=[4, h, w], conditioning=cond,
                          unconditional_guidance_scale=scale,
                          unconditional_conditioning=True if scale != 1.0 else None, eta=eta, x_T=x_T,
                          verbose=False)
    dec = model.decode_first_stage(z)
    img = ops.to_image_u8(dec) if as_uint8 else ops.to_image(dec)
    return img, z


@torch.no_grad()
def img2img(model, init_image, uc, c, strength=0.75, steps=50, scale=5.0, eta=0.0, sampler=None, noise=None,
            posterior_noise=None):
    """The caller-side sequence of img2img.py:269-301 on the device: encode the image, noise it to
    t_enc = int(strength * steps), denoise with CFG, decode.  Returns images in [0,1] and the latents."""
    from . import ops
    from .ddim import DDIMSampler
    sampler = sampler or DDIMSampler(model)
    b = init_image.shape[0]
    init_latent = model.get_first_stage_encoding(model.encode_first_stage(init_image), noise=posterior_noise)
    sampler.make_schedule(ddim_num_steps=steps, ddim_eta=eta, verbose=False)
    assert 0. <= strength <= 1., 'can only work with strength in [0.0, 1.0]'
    t_enc = int(strength * steps)
    z_enc = sampler.stochastic_encode(init_latent, torch.tensor([t_enc] * b, device=init_latent.device), noise=noise)
    cond = torch.cat([uc, c]) if scale != 1.0 else c
    z = sampler.decode(z_enc, cond, t_enc, unconditional_guidance_scale=scale,
                       unconditional_conditioning=True if scale != 1.0 else None)
    return ops.to_image(model.decode_first_stage(z)), z


@torch.no_grad()
def inpaint(model, image, mask, uc, c, steps=50, scale=7.5, eta=0.0, sampler=None, noise=None, posterior_noise=None):
    """inpaint_with_text.py:251-291 on the device (hybrid 9-channel UNet): image (B,3,H,W) in [-1,1], mask
    (B,1,H,W) in {0,1} with 1 = repaint.  Returns the composited image in [0,1] and the latents."""
    from . import ops
    from .ddim import DDIMSampler
    sampler = sampler or DDIMSampler(model)
    b = image.shape[0]
    dev = model.device
    image, mask = image.to(dev, torch.float32), mask.to(dev, torch.float32)
    # make_batch (inpaint_with_text.py:65-70) masks the image in [0,1] and then maps both tensors through
    # * 2 - 1, so repainted pixels become -1: ((1 - m) * (image + 1) / 2) * 2 - 1 = (1 - m) * image - m
    masked_image = (1 - mask) * image - mask
    pn = posterior_noise if posterior_noise is not None else (None, None)
    img_latent = model.get_first_stage_encoding(model.encode_first_stage(image), noise=pn[0])
    masked_latent = model.get_first_stage_encoding(model.encode_first_stage(masked_image), noise=pn[1])
    masked_latent = torch.cat([masked_latent] * 2)
    img_mask = torch.nn.functional.interpolate(mask, size=img_latent.shape[-2:])
    img_mask = torch.cat([img_mask] * 2)
    sampler.make_schedule(ddim_num_steps=steps, ddim_eta=eta, verbose=False)
    x_T = sampler.stochastic_encode(img_latent, torch.tensor([steps - 1], device=dev), noise=noise)
    cond = {'c_concat': [[img_mask], [masked_latent]], 'c_crossattn': [torch.cat([uc, c])]}
    z, _ = sampler.sample(S=steps, batch_size=b, shape=list(img_latent.shape[1:]), conditioning=cond, eta=eta,
                          verbose=False, unconditional_guidance_scale=scale, unconditional_conditioning=True,
                          x_T=x_T)
    predicted = ops.to_image(model.decode_first_stage(z))
    return ops.inpaint_composite(image, mask, predicted), z


def shard(total, rank, world):
    """Contiguous block [lo, hi) of `total` samples owned by `rank`."""
    per = (total + world - 1) // world
    lo = min(total, rank * per)
    return lo, min(total, lo + per)


def sample_batches(lo, hi, batch):
    """[(first global sample index, count)] for the block [lo, hi) walked `batch` samples at a time, the way
    txt2img.py:258-303 walks its prompt list in chunks of n_samples (the last chunk may be short)."""
    return [(s, min(batch, hi - s)) for s in range(lo, hi, batch)]


def run_sample_list(total, batch, run_batch, rank=None, world=None):
    """A fixed list of `total` samples over the ranks of the process group: this rank takes the contiguous
    block ``shard(total, rank, world)``, hands it to ``run_batch(first, count) -> (count, ...) tensor`` in
    chunks of `batch`, and the per-rank results are all-gathered into a (total, ...) tensor that every
    rank receives, in global sample order.  ``run_batch`` must derive everything from the global sample
    index, so the result does not depend on the number of ranks."""
    import torch.distributed as dist
    live = dist.is_available() and dist.is_initialized()
    if world is None:
        world = dist.get_world_size() if live else 1
    if rank is None:
        rank = dist.get_rank() if live else 0
    lo, hi = shard(total, rank, world)
    parts = [run_batch(first, n) for first, n in sample_batches(lo, hi, batch)]
    return gather_sharded(parts, total, rank, world)


def gather_sharded(parts, total, rank, world):
    """Concatenate this rank's batch results and all-gather the (possibly ragged: total need not divide by
    world, trailing ranks may own nothing) per-rank blocks into one (total, ...) tensor."""
    per = (total + world - 1) // world
    lo, hi = shard(total, rank, world)
    if world == 1:
        return torch.cat(parts) if len(parts) != 1 else parts[0]
    import torch.distributed as dist
    # every rank contributes `per` rows (a short last block is zero-padded; the padding lands behind row `total`)
    mine = (torch.cat(parts) if len(parts) != 1 else parts[0]) if parts else None
    if (world - 1) * per >= total:
        # some rank owns no sample and cannot know the row shape: rank 0 (never empty) announces it
        kinds = (torch.float32, torch.uint8, torch.float16)
        meta = [list(mine.shape[1:]) + [kinds.index(mine.dtype), int(mine.is_cuda)] if rank == 0 else None]
        dist.broadcast_object_list(meta, src=0)
        tail, dtype, on_gpu = tuple(meta[0][:-2]), kinds[meta[0][-2]], meta[0][-1]
        dev = torch.device("cuda", torch.cuda.current_device()) if on_gpu else torch.device("cpu")
    else:
        tail, dtype, dev = tuple(mine.shape[1:]), mine.dtype, mine.device
    if mine is not None and hi - lo == per:
        block = mine
    else:
        block = torch.zeros((per,) + tail, dtype=dtype, device=dev)
        if mine is not None:
            block[:hi - lo] = mine
    return gather_images(block, world)[:total]


def gather_images(img, world=None):
    """All-gather equally sized per-rank image batches over RCCL (identity on a single process)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return img
    world = dist.get_world_size()
    if img.is_cuda and dist.get_backend() == "gloo":
        # gloo has no all-gather on device tensors: stage through the host (CPU rehearsals of the multi-rank path)
        host = gather_images(img.cpu(), world)
        return host.to(img.device)
    out = torch.empty((world * img.shape[0],) + tuple(img.shape[1:]), dtype=img.dtype, device=img.device)
    try:
        dist.all_gather_into_tensor(out, img.contiguous())
    except (RuntimeError, NotImplementedError):       # backends without the flat variant
        dist.all_gather(list(out.chunk(world)), img.contiguous())
    return out
