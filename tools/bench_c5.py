#!/usr/bin/env python3
"""Timing of the inpainting / img2img configurations (not the headline metric): 9-channel hybrid UNet at
96x96 latents (768x768 images), batch 2 = one CFG pair, and the two script-side pipelines end to end."""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pipeline = importlib.import_module("stable-diffusion_amd.pipeline")
synth = importlib.import_module("stable-diffusion_amd.synth")


def sync():
    torch.cuda.synchronize()
    return time.perf_counter()


model = pipeline.build_synthetic_model(9, "hybrid", "fp16", encoder=True)
unet = model.model.diffusion_model
h = 96
x = torch.randn(2, 9, h, h, device="cuda")
ctx = torch.cat(pipeline.synthetic_conditioning(1))
t = torch.full((2,), 981, dtype=torch.int64, device="cuda")
o = torch.empty(2, 4, h, h, device="cuda")
for _ in range(3):
    unet(x, t, context=ctx, out=o)
t0 = sync()
for _ in range(20):
    unet(x, t, context=ctx, out=o)
dt = (sync() - t0) / 20
print(f"UNet 9-channel, latent 96x96, batch 2: {dt * 1e3:.2f} ms  ({4296.2 / dt / 1e3:.0f} TFLOP/s algorithmic)")
uc, c = pipeline.synthetic_conditioning(1)
img = synth.uniform(5, "c5.image", (1, 3, 768, 768), -1.0, 1.0).cuda()
mask = (synth.uniform(6, "c5.mask", (1, 1, 768, 768), 0.0, 1.0) > 0.5).float().cuda()
for name, fn in (("inpaint 768x768, 50 steps, CFG", lambda: pipeline.inpaint(model, img, mask, uc, c, steps=50)),):
    fn()
    t0 = sync()
    fn()
    print(f"{name}: {sync() - t0:.3f} s per image")
del model
model = pipeline.build_synthetic_model(4, "crossattn", "fp16", encoder=True)
img512 = img[:, :, :512, :512].contiguous()
fn = lambda: pipeline.img2img(model, img512, uc, c, strength=0.75, steps=50, scale=5.0)
fn()
t0 = sync()
fn()
print(f"img2img 512x512, strength 0.75 (37 UNet steps), CFG: {sync() - t0:.3f} s per image")
