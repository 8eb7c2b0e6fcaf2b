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
    return {k: torch.from_numpy(np.asarray(z[k])) for k in z.files}


def max_rel_err(a, b):
    """max |a-b| / max |b|  -- the per-call error measure used for the UNet parity bar."""
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def psnr(a, b, peak=1.0):
    mse = float(((a.double() - b.double()) ** 2).mean())
    return 99.0 if mse == 0 else 10.0 * float(np.log10(peak * peak / mse))


def g2_u(key, shape, lo=-2., hi=2., device="cpu"):
    return synth.uniform(7, "g2." + key, shape, lo, hi, device)


def sub_state_dict(prefix, shapes_or_keys, seed=0, device="cpu"):
    """Synthesised tensors for keys given as {relative_key: shape}, stored under prefix+key hashes."""
    return {k: synth.synth_tensor(seed, prefix + k, s, device) for k, s in shapes_or_keys.items()}
