#!/usr/bin/env python3
"""Two ranks on one GPU over gloo: which stage of the multi-rank bench path stalls?  (rehearsal aid)"""
import os, sys, time
import torch
import torch.distributed as dist
rank = int(os.environ["RANK"])
def say(msg):
    print(f"[rank {rank} +{time.time() - t0:6.1f}s] {msg}", file=sys.stderr, flush=True)
t0 = time.time()
torch.cuda.set_device(0)
dist.init_process_group("gloo")
say("group up")
x = torch.ones(4, device="cuda") * (rank + 1)
torch.cuda.synchronize()
say(f"device tensor ok {x.sum().item()}")
dist.barrier()
say("barrier ok")
t = torch.tensor([float(rank)], dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
say(f"all_reduce ok {t.item()}")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib
pipeline = importlib.import_module("stable-diffusion_amd.pipeline")
g = pipeline.gather_images(torch.full((2, 3, 8, 8), float(rank), device="cuda"))
say(f"gather ok {tuple(g.shape)} {g[:, 0, 0, 0].tolist()}")
model = pipeline.build_synthetic_model(4, "crossattn", "fp16", device=torch.device("cuda", 0))
say("model built")
x_T = pipeline.initial_latents(2, 16, 16)
uc, c = pipeline.synthetic_conditioning(2)
img, _ = pipeline.txt2img(model, x_T, uc, c, steps=3)
torch.cuda.synchronize()
say(f"txt2img ok {tuple(img.shape)}")
dist.barrier()
dist.destroy_process_group()
say("done")
