#!/usr/bin/env python3
"""A/B of the two matrix-core attention kernels (software-pipelined vs plain) on the UNet's self-attention
shapes, in one process: time per call, TFLOP/s, and the largest difference between their outputs and against an
fp32 torch reference on a slice."""
import importlib, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
ops = importlib.import_module("stable-diffusion_amd.ops")

def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

shapes = [(1, 8, 4096, 40), (2, 8, 4096, 40), (3, 8, 4096, 40), (6, 8, 4096, 40), (8, 8, 4096, 40), (12, 8, 4096, 40), (16, 8, 4096, 40), (16, 8, 1024, 80), (16, 8, 256, 160), (4, 8, 4096, 40), (2, 8, 9216, 40), (16, 8, 4096, 64),
          (16, 8, 1024, 128)]
for dt in (torch.float16,):
    for B, H, N, d in shapes:
        C = H * d
        g = torch.Generator(device="cuda").manual_seed(1)
        qkv = (torch.randn(B, N, 3 * C, device="cuda", generator=g) * 1.0).to(dt)
        q, k, v = qkv[:, :, :C], qkv[:, :, C:2 * C], qkv[:, :, 2 * C:]
        y1 = ops.attention(q, k, v, H, impl=0)
        y0 = ops.attention(q, k, v, H, impl=4)
        # fp32 reference on batch 0, head 0..1
        qq, kk, vv = (t[0].float().reshape(N, H, d).transpose(0, 1)[:2] for t in (q, k, v))
        ref = torch.softmax(qq @ kk.transpose(1, 2) * d ** -0.5, -1) @ vv
        got1 = y1[0].float().reshape(N, H, d).transpose(0, 1)[:2]
        got0 = y0[0].float().reshape(N, H, d).transpose(0, 1)[:2]
        e1 = (got1 - ref).abs().max().item() / ref.abs().max().item()
        e0 = (got0 - ref).abs().max().item() / ref.abs().max().item()
        t1 = timeit(lambda: ops.attention(q, k, v, H, impl=0))
        t0 = timeit(lambda: ops.attention(q, k, v, H, impl=4))
        fl = 4.0 * B * H * N * N * d
        print(f"{str(dt)[6:]:9s} B{B} h{H} N{N} d{d}: pipelined {t1:8.1f} us {fl / t1 / 1e6:7.1f} TF/s err {e1:.2e} | plain {t0:8.1f} us "
              f"{fl / t0 / 1e6:7.1f} TF/s err {e0:.2e} | diff {(y1.float() - y0.float()).abs().max().item():.2e}", flush=True)
