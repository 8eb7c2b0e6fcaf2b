"""GPU: the multi-rank paths of bench.py end to end.  The test box has one GPU, so the ranks share cuda:0 and
talk over gloo (SDMI_BENCH_SHARE_DEVICE / SDMI_BENCH_BACKEND): rendezvous, sharded samples, barrier +
max-over-ranks timing, image gather, rank-0-only roofline pass, one JSON line."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--steps", "1", "--warmup", "1", "--size", "256", "--ddim-steps", "4", "--batch", "2",
         "--no-cpu-baseline", "--no-c2"]


def _one_json_line(res, only=False):
    """The one JSON line of a run; with ``only`` nothing else may be on stdout (under torchrun the gloo
    transport prints its connection notes there, which the launch through `--gpus` moves to stderr)."""
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.strip()]
    if only:
        assert len(lines) == 1, res.stdout
    lines = [l for l in lines if l.startswith("{")]
    assert len(lines) == 1, res.stdout
    return json.loads(lines[0])


def test_bench_two_ranks_one_device():
    """Launched the way the driver launches N > 1: torch.distributed.run around bench.py (weak scaling)."""
    env = dict(os.environ, SDMI_BENCH_BACKEND="gloo", SDMI_BENCH_SHARE_DEVICE="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29541", os.path.join(ROOT, "bench.py"),
           "--gpus", "2"] + SMALL
    res = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=280)
    out = _one_json_line(res)
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["config"]["global_batch"] == 4
    assert out["value"] > 0 and out["steps"] == 1 and out["warmup"] == 1
    assert "roofline" in out and out["unit"] == "images/sec"
    # what lets a scaling record prove that the gather really spanned the ranks: backend and rank count as the
    # process group reports them, the per-rank payload (2 uint8 images of 256x256x3) and the gather's own time
    col = out["collective"]
    assert col["backend"] == "gloo" and col["ranks"] == 2 and col["bytes_per_rank"] == 2 * 256 * 256 * 3
    assert col["ms"] > 0


def test_bench_gpus_flag_starts_the_ranks_and_fixed_list_mode():
    """`python bench.py --gpus 2` with no torchrun and no rank environment: the parent starts the two ranks as a
    child process and relays the one JSON line; --total-images runs the fixed sample list (6 samples in batches of
    2 over 2 ranks: blocks of 3, so each rank runs a full and a short batch)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(SDMI_BENCH_BACKEND="gloo", SDMI_BENCH_SHARE_DEVICE="1")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--total-images", "6"] + SMALL
    res = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=280)
    out = _one_json_line(res, only=True)
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["config"]["global_batch"] == 6
    assert out["value"] > 0 and abs(out["value"] - 6 / (out["ms_per_step"] * 1e-3)) < 1e-6 * out["value"]
    assert out["collective"]["ranks"] == 2 and out["collective"]["bytes_per_rank"] == 3 * 256 * 256 * 3
