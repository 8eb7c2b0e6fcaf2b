#!/usr/bin/env python3
"""One pointwise GEMM launched a few times (for rocprofv3 --pmc passes): gemm_pmc_one.py M N K [resid]"""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ops = importlib.import_module("stable-diffusion_amd.ops")
M, N, K = (int(a) for a in sys.argv[1:4])
resid = len(sys.argv) > 4 and sys.argv[4] == "1"
x = torch.randn(1, 1, M, K, device="cuda").half()
w = (torch.randn(N, K, device="cuda") * 0.02).half()
b = torch.randn(N, device="cuda")
r = torch.randn(1, 1, M, N, device="cuda").half() if resid else None
for _ in range(5):
    ops.conv2d(x, w, b, taps=1, splits=1, resid=r)
torch.cuda.synchronize()
print("ok")
