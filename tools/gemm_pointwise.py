#!/usr/bin/env python3
"""Pointwise (token) GEMMs of the transformer blocks at UNet batch 16 under every tile shape the planner can be
forced to (SDMI_GEMM_SHAPE) and a few split counts: microseconds per launch."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
ops = importlib.import_module("stable-diffusion_amd.ops")
_lib = importlib.import_module("stable-diffusion_amd._lib")
from bench_shapes import timeit

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
shapes = []
for (H, d) in ((64, 320), (32, 640), (16, 1280)):
    M = B * H * H
    shapes += [(M, d, d, 0), (M, 3 * d, d, 0), (M, 8 * d, d, 1), (M, d, 4 * d, 0)]
dt = torch.float16
for (M, N, K, geglu) in shapes:
    x = torch.randn(1, 1, M, K, device="cuda").to(dt)
    w = (torch.randn(N, K, device="cuda") * 0.02).to(dt)
    bias = torch.randn(N, device="cuda")
    res = torch.randn(1, 1, M, N, device="cuda").to(dt) if not geglu else None
    line = f"M{M} N{N} K{K} g{geglu}:"
    for shape in ("0", "1", "2", "3", "4"):
        os.environ["SDMI_GEMM_SHAPE"] = shape
        auto = _lib.lib().sdmi_conv2d_splits(M, N, K, _lib.dtype_code(dt))
        best = None
        for s in sorted({auto, 1, 2}):
            if K // 64 // s < 2:
                continue
            try:
                t = timeit(lambda: ops.conv2d(x, w, bias, taps=1, geglu=bool(geglu), splits=s, resid=res), iters=10) * 1e6
            except Exception as e:
                t = float("nan")
            tag = f"s{s}{'*' if s == auto else ''}={t:.1f}"
            best = tag if best is None else best + " " + tag
        line += f" | shape{shape}: {best}"
    print(line, flush=True)
