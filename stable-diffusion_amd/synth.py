"""Deterministic synthetic weights and inputs.

No checkpoint of the reference exists offline, so every parity fixture and every benchmark uses
SD-v1-shaped tensors drawn from a counter-based generator: element ``i`` of the tensor stored under
checkpoint key ``k`` is a pure function of ``(seed, k, i)``.  The generator is a 32-bit integer
hash evaluated with int64 tensor arithmetic (no overflow-dependent behaviour), so CPU and GPU
produce bit-identical floats and a tensor can be regenerated anywhere without shipping it.

Initialisation rule (torch's own defaults for Conv/Linear, restated): weights and biases
~ U(-1/sqrt(fan_in), 1/sqrt(fan_in)); normalisation gains 1 + U(-0.1, 0.1), shifts U(-0.1, 0.1)
(a non-trivial affine so that parity tests exercise gamma/beta); tensors the reference zero-
initialises (zero_module, stable_diffusion.py:24, used at :3188, :2883, :3495) are re-randomised
like any other weight, otherwise a random-weight UNet returns exactly zero.
"""
import math
import zlib

import torch

_M32 = 0xFFFFFFFF


def _mix32(x: torch.Tensor) -> torch.Tensor:
    """lowbias32-style avalanche on values held in int64, kept within 32 bits."""
    x = x & _M32
    x = x ^ (x >> 16)
    x = (x * 0x7FEB352D) & _M32
    x = x ^ (x >> 15)
    x = (x * 0x846CA68B) & _M32
    x = x ^ (x >> 16)
    return x


def key_hash(seed: int, key: str) -> int:
    return (zlib.crc32(key.encode("utf-8")) ^ (seed * 0x9E3779B1)) & _M32


def uniform01(seed: int, key: str, numel: int, device="cpu", stream: int = 0) -> torch.Tensor:
    """numel floats in [0, 1), fp32, exactly representable (24-bit mantissa draw)."""
    kh = key_hash(seed, key)
    idx = torch.arange(numel, dtype=torch.int64, device=device)
    a = _mix32(idx ^ kh)
    b = _mix32((a + 0x632BE5AB * (stream + 1) + (idx >> 32)) & _M32)
    return (b >> 8).to(torch.float32) * (1.0 / 16777216.0)


def uniform(seed, key, shape, lo, hi, device="cpu") -> torch.Tensor:
    n = int(math.prod(shape)) if len(shape) else 1
    u = uniform01(seed, key, n, device)
    return (lo + (hi - lo) * u).reshape(shape)


def normal(seed, key, shape, device="cpu") -> torch.Tensor:
    """Standard normal via Box-Muller on two hash streams (fp32 math, deterministic on CPU;
    used for latents / contexts, which are always generated on the host and then uploaded)."""
    n = int(math.prod(shape))
    u1 = uniform01(seed, key, n, "cpu", stream=0).double()
    u2 = uniform01(seed, key, n, "cpu", stream=1).double()
    r = torch.sqrt(-2.0 * torch.log(1.0 - u1))
    z = r * torch.cos(2.0 * math.pi * u2)
    return z.to(torch.float32).reshape(shape).to(device)


def synth_tensor(seed: int, key: str, shape, device="cpu") -> torch.Tensor:
    """One checkpoint tensor under the initialisation rule in the module docstring."""
    shape = tuple(shape)
    leaf = key.rsplit(".", 1)[-1]
    is_norm = len(shape) == 1 and _is_norm_key(key)
    if is_norm:
        if leaf == "weight":
            return uniform(seed, key, shape, 0.9, 1.1, device)
        return uniform(seed, key, shape, -0.1, 0.1, device)
    if leaf == "weight":
        fan_in = int(math.prod(shape[1:]))
        bound = 1.0 / math.sqrt(fan_in)
        return uniform(seed, key, shape, -bound, bound, device)
    # bias of a conv/linear: torch bounds it by the fan_in of its weight, which the caller does not
    # pass; a fixed small bound keeps it simple and deterministic.
    return uniform(seed, key, shape, -0.05, 0.05, device)


def _is_norm_key(key: str) -> bool:
    parts = key.split(".")
    name = parts[-2]
    if name.startswith("norm") or "layer_norm" in name:   # norm, norm1..3, norm_out; the text transformer's
        return True
    # GroupNorm32 inside nn.Sequential: in_layers.0 / out_layers.0 / out.0
    if name == "0" and parts[-3] in ("in_layers", "out_layers", "out"):
        return True
    return False


def synth_state_dict(shapes, seed: int = 0, device="cpu", prefix: str = "") -> dict:
    return {prefix + k: synth_tensor(seed, k, s, device) for k, s in shapes.items()}
