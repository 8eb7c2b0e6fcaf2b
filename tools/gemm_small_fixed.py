#!/usr/bin/env python3
"""Fixed (K-independent) cost of the GEMM launches of a batch-2 UNet forward: pointwise problems at growing K, device
time from the library's own event brackets (Python's call overhead exceeds these kernels).  The bracket itself costs
about 4.5 us, printed as measured.  usage: gemm_small_fixed.py [SDMI_GEMM_SHAPE values, default "0 1 4"]"""
import importlib, os, sys, tempfile
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ops = importlib.import_module("stable-diffusion_amd.ops")
_lib = importlib.import_module("stable-diffusion_amd._lib")
lib = _lib.lib()


def device_us(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    lib.sdmi_prof_reset(); lib.sdmi_prof_enable(1)
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    lib.sdmi_prof_enable(0)
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "l.csv")
        lib.sdmi_prof_dump(path.encode())
        rows = [l.split(",") for l in open(path).read().strip().split("\n")[1:]]
    return sum(float(r[2]) for r in rows) / iters, rows[0][1]


shapes = sys.argv[1:] or ["0", "1", "4"]
for (M, N) in ((8192, 320), (2048, 640), (512, 1280), (128, 1280)):
    for K in (64, 128, 320, 640, 1280, 2560, 5120):
        x = torch.randn(1, 1, M, K, device="cuda").half()
        w = (torch.randn(N, K, device="cuda") * 0.02).half()
        b = torch.randn(N, device="cuda")
        r = torch.randn(1, 1, M, N, device="cuda").half()
        line = f"M{M} N{N} K{K:5d}:"
        for sh in shapes:
            os.environ["SDMI_GEMM_SHAPE"] = sh
            s = lib.sdmi_conv2d_splits(M, N, K, _lib.dtype_code(torch.float16))
            t, label = device_us(lambda: ops.conv2d(x, w, b, taps=1, splits=s, resid=r))
            line += f" | shape{sh}: {t:6.1f} us ({label.split(' ', 3)[-1]})"
        print(line, flush=True)
