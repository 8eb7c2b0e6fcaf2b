#!/usr/bin/env python3
"""Timing of the tiny-M linear (embedding projections) on the GPU box."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from bench_shapes import timeit
ops = importlib.import_module("stable-diffusion_amd.ops")
for (M, K, N) in ((16, 320, 1280), (16, 1280, 1280), (16, 1280, 20160), (2, 1280, 20160), (8, 1280, 20160)):
    x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") * 0.02; b = torch.randn(N, device="cuda")
    t = timeit(lambda: ops.small_linear(x, w, b, 1, 0), iters=30)
    print(f"M={M:3d} K={K:5d} N={N:6d}: {t*1e6:8.1f} us  weights {N*K*4/t/1e9:8.1f} GB/s")
