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


def build_synthetic_model(in_channels=4, conditioning_key="crossattn", dtype="fp16", seed=0, device="cuda"):
    """LatentDiffusion with SD-v1-shaped weights from the deterministic synthesiser (no checkpoint
    exists offline).  Hash keys are the un-prefixed checkpoint names."""
    from .ddpm import LatentDiffusion
    params = sd_v1_model_params(in_channels, conditioning_key)
    model = LatentDiffusion(**params, compute_dtype=dtype, device=device)
    sd = {"model.diffusion_model." + k: synth.synth_tensor(seed, k, s, device)
          for k, s in model.model.diffusion_model.state_dict_shapes().items()}
    sd.update({"first_stage_model." + k: synth.synth_tensor(seed, k, s, device)
               for k, s in model.first_stage_model.state_dict_shapes().items()})
    model.load_state_dict(sd)
    return model


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
def txt2img(model, x_T, uc, c, steps=50, scale=7.5, eta=0.0, sampler=None):
    """One batch through the whole path; returns images (B,3,8h,8w) in [0,1] and the final latents."""
    from . import ops
    from .ddim import DDIMSampler
    sampler = sampler or DDIMSampler(model)
    b, _, h, w = x_T.shape
    cond = torch.cat([uc, c]) if scale != 1.0 else c
    z, _ = sampler.sample(S=steps, batch_size=b, shape=[4, h, w], conditioning=cond,
                          unconditional_guidance_scale=scale,
                          unconditional_conditioning=True if scale != 1.0 else None, eta=eta, x_T=x_T,
                          verbose=False)
    img = ops.to_image(model.decode_first_stage(z))
    return img, z


def shard(total, rank, world):
    """Contiguous block [lo, hi) of `total` samples owned by `rank`."""
    per = (total + world - 1) // world
    lo = min(total, rank * per)
    return lo, min(total, lo + per)


def gather_images(img, world=None):
    """All-gather equally sized per-rank image batches over RCCL (identity on a single process)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return img
    world = dist.get_world_size()
    out = torch.empty((world * img.shape[0],) + tuple(img.shape[1:]), dtype=img.dtype, device=img.device)
    try:
        dist.all_gather_into_tensor(out, img.contiguous())
    except (RuntimeError, NotImplementedError):       # backends without the flat variant
        dist.all_gather(list(out.chunk(world)), img.contiguous())
    return out
