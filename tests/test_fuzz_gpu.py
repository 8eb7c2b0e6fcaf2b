"""Randomised shapes and kernel choices for the C-ABI operators (tools/fuzz_ops.py) as part of the GPU suite."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [11, 12])
def test_operator_fuzz(seed):
    """300 random cases per operator and seed (attention, convolution, GroupNorm, LayerNorm, LayerNorm-linear) against
    fp64 torch arithmetic on the same device; tile shape, K order, store path and the short-key attention walk are
    forced at random.  The tool exits non-zero and lists the cases when anything is outside its tolerance."""
    env = dict(os.environ)
    for name in ("SDMI_GEMM_SHAPE", "SDMI_GEMM_TAP_INNER", "SDMI_GEMM_DIRECT_STORE", "SDMI_XATTN_QPW"):
        env.pop(name, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_ops.py"), str(seed), "300"], env=env,
                       capture_output=True, text=True, timeout=110)
    print(r.stdout[-3000:])
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
