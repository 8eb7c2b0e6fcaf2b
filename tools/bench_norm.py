#!/usr/bin/env python3
"""Timing of LayerNorm / GroupNorm shapes of the UNet against a plain device copy (GPU box)."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from bench_shapes import timeit
ops = importlib.import_module("stable-diffusion_amd.ops")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
for (hw, d) in ((4096, 320), (1024, 640), (256, 1280), (64, 1280)):
    rows = B * hw
    x = torch.randn(rows, d, device="cuda").half()
    g = torch.rand(d, device="cuda") + 0.5; b = torch.randn(d, device="cuda")
    y = torch.empty_like(x)
    tc = timeit(lambda: y.copy_(x), iters=50)
    tl = timeit(lambda: ops.layernorm(x, g, b), iters=50)
    x4 = x.view(B, hw, 1, d)
    tg = timeit(lambda: ops.groupnorm(x4, g, b, 1e-5, True), iters=50)
    mb = 2 * rows * d * 2 / 1e6
    print(f"rows {rows:6d} d {d:5d} ({mb:6.1f} MB r+w): copy {tc*1e6:6.1f} us {mb/tc/1e6:6.2f} TB/s | "
          f"layernorm {tl*1e6:6.1f} us {mb/tl/1e6:6.2f} TB/s | groupnorm+silu {tg*1e6:6.1f} us {mb/tg/1e6:6.2f} TB/s")
