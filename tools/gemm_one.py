#!/usr/bin/env python3
"""Time one pointwise GEMM shape under each forced kernel choice: gemm_one.py M N K [splits]
(SDMI_GEMM_SHAPE is read per call: 0 planner, 1 128x128 kernel, 2 256x320/256x256, 3 256x160, 4 128x160,
5 planner without 128x160)."""
import importlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
from bench_shapes import timeit
ops = importlib.import_module("stable-diffusion_amd.ops")
M, N, K = (int(a) for a in sys.argv[1:4])
x = torch.randn(1, 1, M, K, device="cuda").half()
w = (torch.randn(N, K, device="cuda") * 0.02).half()
b = torch.randn(N, device="cuda")
r = torch.randn(1, 1, M, N, device="cuda").half()
for shape in ("0", "1", "2", "3", "4", "5"):
    os.environ["SDMI_GEMM_SHAPE"] = shape
    for splits in ([None] if shape in ("0", "5") else [1, 2, 4]):
        try:
            t = timeit(lambda: ops.conv2d(x, w, b, taps=1, splits=splits, resid=r), iters=50)
        except RuntimeError as e:
            print(f"shape {shape} splits {splits}: {str(e)[:80]}")
            continue
        print(f"shape {shape} splits {splits}: {t*1e6:7.1f} us  {2.0*M*N*K/t/1e12:6.1f} TF/s")
# which plan did the library choose?
import tempfile
_lib = importlib.import_module("stable-diffusion_amd._lib")
lib = _lib.lib()
os.environ["SDMI_GEMM_SHAPE"] = "0"
lib.sdmi_prof_reset(); lib.sdmi_prof_enable(1)
ops.conv2d(x, w, b, taps=1, splits=None, resid=r)
torch.cuda.synchronize()
lib.sdmi_prof_enable(0)
with tempfile.TemporaryDirectory() as d:
    path = os.path.join(d, "l.csv")
    lib.sdmi_prof_dump(path.encode())
    print(open(path).read().strip())
