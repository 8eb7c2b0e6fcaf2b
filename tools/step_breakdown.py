#!/usr/bin/env python3
"""Where one bench step (8 images, 50 DDIM steps) spends its wall time (run on the GPU box)."""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pipeline = importlib.import_module("stable-diffusion_amd.pipeline")
sdmi = importlib.import_module("stable-diffusion_amd")
ops = importlib.import_module("stable-diffusion_amd.ops")

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
model = pipeline.build_synthetic_model(4, "crossattn", "fp16")
unet = model.model.diffusion_model
sampler = sdmi.DDIMSampler(model)
uc, c = pipeline.synthetic_conditioning(B)
x_T = pipeline.initial_latents(B, 64, 64)
cond = torch.cat([uc, c])


def sync():
    torch.cuda.synchronize()
    return time.perf_counter()


def sample():
    return sampler.sample(S=50, batch_size=B, shape=[4, 64, 64], conditioning=cond, unconditional_guidance_scale=7.5,
                          unconditional_conditioning=True, eta=0.0, x_T=x_T, verbose=False)[0]


for _ in range(2):
    z = sample()
    img = ops.to_image(model.decode_first_stage(z))
t0 = sync()
z = sample()
t1 = sync()
d = model.decode_first_stage(z)
t2 = sync()
img = ops.to_image(d)
t3 = sync()
print(f"sampler.sample (50 steps): {1e3 * (t1 - t0):8.2f} ms  ({20 * (t1 - t0):.3f} ms/step)")
print(f"decode_first_stage:        {1e3 * (t2 - t1):8.2f} ms")
print(f"to_image:                  {1e3 * (t3 - t2):8.2f} ms")
# bare UNet graph replay at batch 2B
x = torch.cat([x_T, x_T]); t = torch.full((2 * B,), 981, dtype=torch.int64, device="cuda"); o = torch.empty_like(x)
for _ in range(3):
    unet(x, t, context=cond, out=o)
t0 = sync()
for _ in range(20):
    unet(x, t, context=cond, out=o)
t1 = sync()
print(f"UNet forward batch {2 * B} (graph replay): {1e3 * (t1 - t0) / 20:.3f} ms")
unet.set_graph_mode(False)
for _ in range(3):
    unet(x, t, context=cond, out=o)
t0 = sync()
for _ in range(20):
    unet(x, t, context=cond, out=o)
t1 = sync()
print(f"UNet forward batch {2 * B} (eager):        {1e3 * (t1 - t0) / 20:.3f} ms")
