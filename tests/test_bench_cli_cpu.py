"""CPU: the launch logic of bench.py that runs before any GPU call."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env):
    return subprocess.run([sys.executable, BENCH] + args, cwd=ROOT, env=env, stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, text=True, timeout=120)


def test_bench_rejects_a_world_size_that_contradicts_gpus():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    res = _run(["--gpus", "2", "--steps", "1"], env)
    assert res.returncode == 2 and "does not match WORLD_SIZE" in res.stderr and res.stdout == ""


def test_bench_gpus_flag_needs_that_many_devices():
    """Outside torchrun `--gpus N` starts N ranks itself; with fewer visible devices it says so instead of
    starting ranks that would fail one by one (devices are counted from sysfs and the environment, with no HIP call)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "SDMI_BENCH_SHARE_DEVICE")}
    env["HIP_VISIBLE_DEVICES"] = ""
    res = _run(["--gpus", "64", "--steps", "1"], env)
    assert res.returncode == 2 and "device(s) are visible" in res.stderr and res.stdout == ""
