#!/usr/bin/env python3
"""Time ops.attention on the UNet's attention shapes (library from SDMI_LIB): one line of microseconds."""
import importlib, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
ops = importlib.import_module("stable-diffusion_amd.ops")
out = []
for B, H, Nq, Nk, d in [(16, 8, 4096, 4096, 40), (16, 8, 1024, 1024, 80), (16, 8, 256, 256, 160), (16, 8, 64, 64, 160),
                        (16, 8, 4096, 77, 40), (16, 8, 1024, 77, 80), (2, 8, 4096, 4096, 40), (2, 8, 9216, 9216, 40)]:
    C = H * d
    g = torch.Generator(device="cuda").manual_seed(1)
    q = torch.randn(B, Nq, C, device="cuda", generator=g).half()
    k = torch.randn(B, Nk, C, device="cuda", generator=g).half()
    v = torch.randn(B, Nk, C, device="cuda", generator=g).half()
    f = lambda: ops.attention(q, k, v, H)
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        f()
    e1.record()
    torch.cuda.synchronize()
    out.append(f"B{B}q{Nq}k{Nk}d{d} {e0.elapsed_time(e1) / 20 * 1e3:6.1f}")
print(os.path.basename(os.environ.get("SDMI_LIB", "default")), " | ".join(out), flush=True)
