#!/usr/bin/env python3
"""Per-shape timing of the hot kernels (run on the GPU box): python tools/bench_shapes.py [gemm|gn|attn]"""
import importlib, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ops = importlib.import_module("stable-diffusion_amd.ops")


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


def gemm_shapes(B):
    s = []
    # (name, B, H, W, C0, C1, Cout, taps, stride, up)
    for (H, C) in ((64, 320), (32, 640), (16, 1280), (8, 1280)):
        s.append((f"conv3x3 {C}->{C} @{H}", B, H, H, C, 0, C, 9, 1, 0))
    s.append(("conv3x3 960->320 @64 (concat)", B, 64, 64, 640, 320, 320, 9, 1, 0))
    s.append(("conv3x3 1920->640 @32 (concat)", B, 32, 32, 1280, 640, 640, 9, 1, 0))
    s.append(("conv3x3 2560->1280 @16 (concat)", B, 16, 16, 1280, 1280, 1280, 9, 1, 0))
    s.append(("conv3x3 2560->1280 @8 (concat)", B, 8, 8, 1280, 1280, 1280, 9, 1, 0))
    s.append(("down 320 @64->32", B, 64, 64, 320, 0, 320, 9, 2, 0))
    s.append(("up 640 @32->64", B, 32, 32, 640, 0, 640, 9, 1, 1))
    for (H, d) in ((64, 320), (32, 640), (16, 1280)):
        s.append((f"qkv {d}->{3*d} tok@{H}", 1, 1, B * H * H, d, 0, 3 * d, 1, 1, 0))
        s.append((f"proj {d}->{d} tok@{H}", 1, 1, B * H * H, d, 0, d, 1, 1, 0))
        s.append((f"ff1 {d}->{8*d} tok@{H}", 1, 1, B * H * H, d, 0, 8 * d, 1, 1, 0))
        s.append((f"ff2 {4*d}->{d} tok@{H}", 1, 1, B * H * H, 4 * d, 0, d, 1, 1, 0))
    return s


def run_gemm(B, dt=torch.float16):
    print(f"--- conv_gemm, batch {B}")
    tot_t = tot_f = 0
    for (name, b, H, W, C0, C1, Co, taps, stride, up) in gemm_shapes(B):
        x0 = torch.randn(b, H, W, C0, device="cuda").to(dt)
        x1 = torch.randn(b, H, W, C1, device="cuda").to(dt) if C1 else None
        w = (torch.randn(Co, taps * (C0 + C1), device="cuda") * 0.02).to(dt)
        bias = torch.randn(Co, device="cuda")
        geglu = name.startswith("ff1")
        Hup = 2 * H if up else H
        Ho = (Hup + 2 - 3) // stride + 1 if taps == 9 else H
        Wo = ((2 * W if up else W) + 2 - 3) // stride + 1 if taps == 9 else W
        M = b * Ho * Wo
        K = taps * (C0 + C1)
        _lib = importlib.import_module("stable-diffusion_amd._lib")
        splits = _lib.lib().sdmi_conv2d_splits(M, Co, K, _lib.dtype_code(dt))
        fn = lambda: ops.conv2d(x0, w, bias, taps=taps, stride=stride, up=bool(up), x1=x1, geglu=geglu, splits=splits)
        t = timeit(fn)
        fl = 2.0 * M * Co * K
        tot_t += t; tot_f += fl
        print(f"{name:38s} M={M:6d} N={Co:5d} K={K:5d} splits={splits:2d}  {t*1e6:8.1f} us  {fl/t/1e12:7.1f} TF/s")
    print(f"sum: {tot_t*1e3:.2f} ms, {tot_f/tot_t/1e12:.1f} TF/s")


def run_gn(B, dt=torch.float16):
    print(f"--- groupnorm+silu, batch {B}")
    for (H, C) in ((64, 320), (64, 640), (64, 960), (32, 640), (32, 1920), (16, 1280), (16, 2560), (8, 1280), (8, 2560)):
        x = torch.randn(B, H, H, C, device="cuda").to(dt)
        g = torch.ones(C, device="cuda"); b = torch.zeros(C, device="cuda")
        t = timeit(lambda: ops.groupnorm(x, g, b, 1e-5, silu=True))
        by = 2.0 * x.numel() * x.element_size()
        print(f"GN C={C:5d} @{H:3d}  {t*1e6:8.1f} us  {by/t/1e9:8.1f} GB/s (2 x numel x size)")
    for (H, C) in ((512, 128), (256, 256), (128, 512)):
        x = torch.randn(min(B, 2), H, H, C, device="cuda").to(dt)
        g = torch.ones(C, device="cuda"); b = torch.zeros(C, device="cuda")
        t = timeit(lambda: ops.groupnorm(x, g, b, 1e-6, silu=True))
        by = 2.0 * x.numel() * x.element_size()
        print(f"GN(vae) C={C:5d} @{H:3d} B={x.shape[0]}  {t*1e6:8.1f} us  {by/t/1e9:8.1f} GB/s")


def run_attn(B, dt=torch.float16):
    print(f"--- attention, batch {B}")
    for (N, Nk, d) in ((4096, 4096, 40), (4096, 77, 40), (1024, 1024, 80), (1024, 77, 80), (256, 256, 160), (256, 77, 160), (64, 64, 160)):
        C = 8 * d
        q = torch.randn(B, N, C, device="cuda").to(dt); k = torch.randn(B, Nk, C, device="cuda").to(dt); v = torch.randn(B, Nk, C, device="cuda").to(dt)
        t = timeit(lambda: ops.attention(q, k, v, 8))
        fl = 4.0 * B * 8 * N * Nk * d
        print(f"attn N={N:5d} Nk={Nk:5d} d={d:3d}  {t*1e6:8.1f} us  {fl/t/1e12:7.1f} TF/s")


if __name__ == "__main__":
    what = sys.argv[1:] or ["gemm", "gn", "attn"]
    if "gemm16" in what:
        run_gemm(16)
        sys.exit(0)
    for B in (16, 2):
        if "gemm" in what: run_gemm(B)
        if "gn" in what: run_gn(B)
        if "attn" in what: run_attn(B)
