#!/usr/bin/env python3
"""N eager UNet forwards at a given batch (for rocprofv3 --kernel-trace --stats).  usage: unet_loop.py [batch] [iters] [graph]"""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pipeline = importlib.import_module("stable-diffusion_amd.pipeline")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
model = pipeline.build_synthetic_model(4, "crossattn", "fp16")
unet = model.model.diffusion_model
unet.set_graph_mode(len(sys.argv) > 3 and sys.argv[3] == "graph")
x = pipeline.initial_latents(B, 64, 64)
ctx = torch.cat(pipeline.synthetic_conditioning(B // 2))
t = torch.full((B,), 981, dtype=torch.int64, device="cuda")
o = torch.empty_like(x)
import time
for _ in range(3):
    unet(x, t, context=ctx, out=o)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iters):
    unet(x, t, context=ctx, out=o)
torch.cuda.synchronize()
print(f"ok {1e3 * (time.perf_counter() - t0) / iters:.3f} ms per forward")
