#!/usr/bin/env python3
"""Per-launch profile of one eager UNet forward (run on the GPU box):
    python tools/layer_profile.py [vae] [cfg] [batch] [out.csv]
Prints the launches grouped by (class, shape label), sorted by total time."""
import collections, importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pipeline = importlib.import_module("stable-diffusion_amd.pipeline")
_lib = importlib.import_module("stable-diffusion_amd._lib")


def main():
    vae = len(sys.argv) > 1 and sys.argv[1] == "vae"
    if vae:
        sys.argv.pop(1)
    sampler_mode = len(sys.argv) > 1 and sys.argv[1] == "cfg"     # the promises DDIMSampler makes around its call
    if sampler_mode:
        sys.argv.pop(1)
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", f"layers_{'vae_' if vae else ''}b{B}.csv")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    lib = _lib.lib()
    model = pipeline.build_synthetic_model(4, "crossattn", os.environ.get("SDMI_PROFILE_DTYPE", "fp16"))
    unet = model.model.diffusion_model
    unet.set_graph_mode(False)
    if sampler_mode:
        unet.set_uniform_timesteps(True)
        unet.set_cfg_duplicate(True)
    x = pipeline.initial_latents(B, 64, 64)
    ctx = torch.cat(pipeline.synthetic_conditioning(B // 2)) if B >= 2 else pipeline.synthetic_conditioning(1)[0]
    t = torch.full((B,), 981, dtype=torch.int64, device="cuda")
    o = torch.empty_like(x)
    fwd = (lambda: model.decode_first_stage(x)) if vae else (lambda: unet(x, t, context=ctx, out=o))
    for _ in range(3):
        fwd()
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for _ in range(3):
        fwd()
    torch.cuda.synchronize()
    print(f"wall per forward: {(time.perf_counter() - t0) / 3 * 1e3:.2f} ms")
    lib.sdmi_prof_reset()
    lib.sdmi_prof_enable(1)
    reps = 5
    for _ in range(reps):
        fwd()
    torch.cuda.synchronize()
    lib.sdmi_prof_enable(0)
    _lib.check(lib.sdmi_prof_dump(out.encode()), "prof_dump")
    agg = collections.OrderedDict()
    with open(out) as f:
        next(f)
        for line in f:
            cls, label, us, fl, by = line.rstrip("\n").split(",")
            a = agg.setdefault((cls, label), [0, 0.0, 0.0, 0.0])
            a[0] += 1; a[1] += float(us); a[2] += float(fl); a[3] += float(by)
    tot = sum(a[1] for a in agg.values()) / reps
    print(f"batch {B}: profiled kernel time per forward {tot / 1e3:.3f} ms")
    for (cls, label), a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        n, us = a[0] / reps, a[1] / reps
        rate = f"{a[2] / a[1] / 1e6:7.1f} TF/s" if a[2] > 0 else f"{a[3] / a[1] / 1e3:7.1f} GB/s"
        print(f"{cls:10s} {label:44s} n={n:5.1f} {us:9.1f} us ({100 * us / tot:5.1f}%)  {us / n:8.1f} us/launch  {rate}")


if __name__ == "__main__":
    main()
