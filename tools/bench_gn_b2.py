#!/usr/bin/env python3
"""GroupNorm+SiLU device time at small batch through the library's profiler (GPU box): bench_gn_b2.py [B]"""
import importlib, os, sys, tempfile
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ops = importlib.import_module("stable-diffusion_amd.ops")
_lib = importlib.import_module("stable-diffusion_amd._lib")
lib = _lib.lib()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
for (hw, c) in ((4096, 320), (4096, 640), (4096, 960), (1024, 320), (1024, 640), (1024, 960), (1024, 1280), (1024, 1920),
                (256, 640), (256, 1280), (256, 1920), (256, 2560), (64, 1280), (64, 2560)):
    x = torch.randn(B, hw, 1, c, device="cuda").half()
    g = torch.rand(c, device="cuda") + 0.5; b = torch.randn(c, device="cuda")
    for _ in range(3):
        ops.groupnorm(x, g, b, 1e-5, True)
    torch.cuda.synchronize()
    lib.sdmi_prof_reset(); lib.sdmi_prof_enable(1)
    for _ in range(20):
        ops.groupnorm(x, g, b, 1e-5, True)
    torch.cuda.synchronize()
    lib.sdmi_prof_enable(0)
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "l.csv")
        lib.sdmi_prof_dump(path.encode())
        us = [float(l.split(",")[2]) for l in open(path).read().strip().split("\n")[1:]]
    print(f"B{B} HW{hw} C{c}: {sum(us) / len(us):6.1f} us per GroupNorm (incl. ~4.5 us event bracket)")
