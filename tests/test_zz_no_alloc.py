"""GPU: sdmi_groupnorm_static, the GroupNorm entry point that allocates nothing (its scratch is a device array
built into the library), against sdmi_groupnorm with a caller-provided scratch buffer: same kernels, so the
outputs must be equal bit for bit.  Two calls in a row, the second on a smaller input (the built-in buffer keeps
the first call's contents), through the C ABI."""
import importlib

import pytest
import torch

from _util import synth

pytestmark = pytest.mark.gpu

_lib = importlib.import_module("stable-diffusion_amd._lib")
ops = importlib.import_module("stable-diffusion_amd.ops")


def _static(x, gamma, beta, eps, silu, x1=None, groups=32):
    B, H, W, C0 = x.shape
    C1 = x1.shape[3] if x1 is not None else 0
    out = torch.empty((B, H, W, C0 + C1), dtype=x.dtype, device=x.device)
    rc = _lib.lib().sdmi_groupnorm_static(_lib.ptr(x), C0, _lib.ptr(x1), C1, B, H * W, groups, eps, _lib.ptr(gamma),
                                          _lib.ptr(beta), 1 if silu else 0, _lib.ptr(out), _lib.dtype_code(x.dtype),
                                          _lib.stream_ptr())
    return rc, out


def _case(key, B, H, W, C0, C1, dt):
    x = synth.uniform(9, key + ".x", (B, H, W, C0), -2.0, 2.0).cuda().to(dt)
    x1 = synth.uniform(9, key + ".x1", (B, H, W, C1), -2.0, 2.0).cuda().to(dt) if C1 else None
    gamma = synth.uniform(9, key + ".g", (C0 + C1,), 0.5, 1.5).cuda()
    beta = synth.uniform(9, key + ".b", (C0 + C1,), -0.5, 0.5).cuda()
    return x, x1, gamma, beta


@pytest.mark.parametrize("dt", [torch.float16, torch.float32])
def test_static_scratch_groupnorm_equals_workspace_groupnorm(dt):
    before = torch.cuda.memory_allocated()
    # the first two calls are large enough for the two-launch kernels, which pass their partial sums through the
    # scratch buffer: 4 samples of 128x128x320 (many strips per sample), then a smaller two-source input; the last
    # two are small tensors, which the one-launch kernel serves without touching the buffer
    for key, shape, eps, silu in (("na.big", (4, 128, 128, 320, 0), 1e-5, True), ("na.concat", (2, 64, 64, 640, 320), 1e-6, False),
                                  ("na.mid", (16, 32, 32, 320, 0), 1e-5, True), ("na.small", (2, 8, 8, 640, 320), 1e-6, False)):
        x, x1, gamma, beta = _case(key, *shape, dt)
        ref = ops.groupnorm(x, gamma, beta, eps, silu=silu, x1=x1)
        rc, out = _static(x, gamma, beta, eps, silu, x1=x1)
        torch.cuda.synchronize()
        assert rc == 0, _lib.lib().sdmi_last_error()
        assert torch.equal(out, ref), key
        # and against plain torch, so that two equal wrong answers cannot pass
        xin = torch.cat([x, x1], dim=-1) if x1 is not None else x
        t = torch.nn.functional.group_norm(xin.float().permute(0, 3, 1, 2), 32, gamma, beta, eps)
        t = torch.nn.functional.silu(t) if silu else t
        err = float((out.float().permute(0, 3, 1, 2) - t).abs().max() / t.abs().max())
        assert err < (3e-3 if dt == torch.float16 else 2e-5), (key, err)
        del x, x1, ref, out, xin, t
    assert torch.cuda.memory_allocated() <= before + (1 << 20)


def test_static_scratch_groupnorm_rejects_what_does_not_fit():
    x, _, gamma, beta = _case("na.toobig", 33, 2, 2, 64, 0, torch.float16)
    rc, _ = _static(x, gamma, beta, 1e-5, True)
    assert rc == -22 and b"at most 32 samples" in _lib.lib().sdmi_last_error()
