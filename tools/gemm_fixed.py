#!/usr/bin/env python3
"""Fixed (K-independent) cost of the GEMM kernels: pointwise M=65536 N=320 at growing K."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from bench_shapes import timeit
ops = importlib.import_module("stable-diffusion_amd.ops")
M = 65536
for N in (320, 640):
    for K in (64, 128, 320, 640, 1280, 2560):
        x = torch.randn(1, 1, M, K, device="cuda").half()
        w = (torch.randn(N, K, device="cuda") * 0.02).half()
        b = torch.randn(N, device="cuda")
        r = torch.randn(1, 1, M, N, device="cuda").half()
        t = timeit(lambda: ops.conv2d(x, w, b, taps=1, splits=1), iters=50)
        t2 = timeit(lambda: ops.conv2d(x, w, b, taps=1, splits=1, resid=r), iters=50)
        print(f"N={N} K={K:5d}: {t*1e6:7.1f} us   with resid {t2*1e6:7.1f} us   ({2.0*M*N*K/t/1e12:6.1f} TF/s)")
