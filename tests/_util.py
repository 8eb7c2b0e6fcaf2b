"""Shared helpers for the test-suite."""
import importlib
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")

structure = importlib.import_module("stable-diffusion_amd.structure")
synth = importlib.import_module("stable-diffusion_amd.synth")


def golden(name):
    path = os.path.join(GOLD, name + ".npz")
    z = np.load(path, allow_pickle=False)
    # text entries (e.g. the version of the library that produced a vector) stay python strings
    return {k: (str(z[k]) if z[k].dtype.kind in "US" else torch.from_numpy(np.asarray(z[k]))) for k in z.files}


def rel_l2(a, b):
    """||a-b||_2 / ||b||_2."""
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def worst_elem_err(a, b, floor=1e-2):
    """max over elements of |a-b| / max(|b|, floor * max|b|): the element-wise relative error, with the
    denominator floored at 1 % of the largest reference magnitude so that near-zero entries do not divide by ~0.
    A localised error in a low-magnitude region shows up here up to 100 x larger than in max_rel_err."""
    a, b = a.double(), b.double()
    den = b.abs().clamp_min(floor * float(b.abs().max().clamp_min(1e-30)))
    return float(((a - b).abs() / den).max())


def max_rel_err(a, b):
    """max |a-b| / max |b|  -- the per-call error measure used for the UNet parity bar (one global normaliser).
    Every call also prints the two stricter companions, relative L2 and the worst element-wise error, so a test
    log shows all three (`pytest -s`, or the captured output of a failure)."""
    a, b = a.double(), b.double()
    e = float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
    if a.numel() > 1:
        print(f"  [parity] max|a-b|/max|b| {e:.3e}   rel-L2 {rel_l2(a, b):.3e}   worst element-wise "
              f"(floor 1% of max|b|) {worst_elem_err(a, b):.3e}   n={a.numel()}")
    return e


def psnr(a, b, peak=1.0):
    mse = float(((a.double() - b.double()) ** 2).mean())
    return 99.0 if mse == 0 else 10.0 * float(np.log10(peak * peak / mse))


def g2_u(key, shape, lo=-2., hi=2., device="cpu"):
    return synth.uniform(7, "g2." + key, shape, lo, hi, device)


def sub_state_dict(prefix, shapes_or_keys, seed=0, device="cpu"):
    """Synthesised tensors for keys given as {relative_key: shape}, stored under prefix+key hashes."""
    return {k: synth.synth_tensor(seed, prefix + k, s, device) for k, s in shapes_or_keys.items()}


# ---- device-side model builders (GPU tests, smoke, bench) -------------------------------------------
_models = {}


def device_unet(cin=4, dtype="fp16", seed=0):
    """UNetModel with synthesised SD-v1-shaped weights on cuda:0 (cached per configuration)."""
    key = ("unet", cin, dtype, seed)
    if key not in _models:
        sdmi = importlib.import_module("stable-diffusion_amd")
        cfg = structure.sd_v1_unet_config(cin)
        net = sdmi.UNetModel(**cfg, compute_dtype=dtype)
        net.load_state_dict(synth.synth_state_dict(net.state_dict_shapes(), seed, device="cuda"))
        _models[key] = net
    return _models[key]


def device_latent_diffusion(cin=4, conditioning_key="crossattn", dtype="fp16", seed=0):
    key = ("ldm", cin, conditioning_key, dtype, seed)
    if key not in _models:
        sdmi = importlib.import_module("stable-diffusion_amd")
        params = structure.sd_v1_model_params(cin, conditioning_key)
        model = sdmi.LatentDiffusion(**params, compute_dtype=dtype)
        # hash keys are the un-prefixed names, exactly as the fixtures were generated
        sd = {"model.diffusion_model." + k: synth.synth_tensor(seed, k, s, "cuda")
              for k, s in model.model.diffusion_model.state_dict_shapes().items()}
        sd.update({"first_stage_model." + k: synth.synth_tensor(seed, k, s, "cuda")
                   for k, s in model.first_stage_model.state_dict_shapes(encoder=True).items()})
        model.load_state_dict(sd)
        _models[key] = model
    return _models[key]


def drop_models():
    _models.clear()
    import gc
    gc.collect()
    torch.cuda.empty_cache()
