#!/usr/bin/env python3
"""Run one attention shape a few times (for rocprofv3 passes).  usage: one_attn.py B N Nk d [iters]"""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ops = importlib.import_module("stable-diffusion_amd.ops")
B, N, Nk, d = (int(a) for a in sys.argv[1:5])
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 5
C = 8 * d
q = torch.randn(B, N, C, device="cuda").half(); k = torch.randn(B, Nk, C, device="cuda").half()
v = torch.randn(B, Nk, C, device="cuda").half()
for _ in range(iters):
    y = ops.attention(q, k, v, 8)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    y = ops.attention(q, k, v, 8)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / iters
print(f"ok {float(y.float().abs().mean()):.5f}  {us:.1f} us/launch  {4.0 * B * 8 * N * Nk * d / us / 1e6:.1f} TF/s")
