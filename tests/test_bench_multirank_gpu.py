"""GPU: the multi-rank path of bench.py end to end (two ranks sharing cuda:0 over gloo, since the test box
has one GPU): rendezvous, sharded samples, barrier + max-over-ranks timing, image gather, rank-0-only
roofline pass, one JSON line."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_two_ranks_one_device():
    env = dict(os.environ, SDMI_BENCH_BACKEND="gloo", SDMI_BENCH_SHARE_DEVICE="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29541", os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--steps", "1", "--warmup", "1", "--size", "256", "--ddim-steps", "4", "--batch", "2",
           "--no-cpu-baseline", "--no-c2"]
    res = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=280)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["config"]["global_batch"] == 4
    assert out["value"] > 0 and out["steps"] == 1 and out["warmup"] == 1
    assert "roofline" in out and out["unit"] == "images/sec"
