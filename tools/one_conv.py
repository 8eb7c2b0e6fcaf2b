#!/usr/bin/env python3
"""Run one conv_gemm shape a few times (for rocprofv3 --pmc passes).  usage: one_conv.py B H C0 C1 Cout taps"""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ops = importlib.import_module("stable-diffusion_amd.ops")
B, H, C0, C1, Co, taps = (int(a) for a in sys.argv[1:7])
iters = int(sys.argv[7]) if len(sys.argv) > 7 else 5
dt = torch.float16
x0 = torch.randn(B, H, H, C0, device="cuda").to(dt)
x1 = torch.randn(B, H, H, C1, device="cuda").to(dt) if C1 else None
w = (torch.randn(Co, taps * (C0 + C1), device="cuda") * 0.02).to(dt)
bias = torch.randn(Co, device="cuda")
for _ in range(iters):
    y = ops.conv2d(x0, w, bias, taps=taps, x1=x1)
torch.cuda.synchronize()
print("ok", float(y.float().abs().mean()))
