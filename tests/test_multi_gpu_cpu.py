"""CPU, world_size 2 over gloo: the sample sharding and image gather used by the multi-GPU path."""
import importlib
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from _util import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, total, q):
    import sys
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pipeline = importlib.import_module("stable-diffusion_amd.pipeline")
    lo, hi = pipeline.shard(total, rank, world)
    # stand-in for the decoded images of this rank's samples: a function of the GLOBAL sample index only
    x = pipeline.initial_latents(hi - lo, 4, 4, device="cpu", first_sample=lo)
    img = torch.tanh(x)[:, :3]
    out = pipeline.gather_images(img)
    q.put((rank, lo, hi, out))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_and_gather_world2():
    pipeline = importlib.import_module("stable-diffusion_amd.pipeline")
    total, world = 6, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process result over all samples
    ref = torch.tanh(pipeline.initial_latents(total, 4, 4, device="cpu", first_sample=0))[:, :3]
    ranges = sorted((lo, hi) for _, lo, hi, _ in got)
    assert ranges == [(0, 3), (3, 6)]
    for rank, lo, hi, out in got:
        assert out.shape == ref.shape
        assert torch.equal(out, ref), f"rank {rank}: gathered images differ from the unsharded run"


def test_shard_covers_everything_once():
    pipeline = importlib.import_module("stable-diffusion_amd.pipeline")
    for total in (1, 7, 8, 64):
        for world in (1, 2, 4, 8):
            seen = []
            for r in range(world):
                lo, hi = pipeline.shard(total, r, world)
                seen.extend(range(lo, hi))
            assert seen == list(range(total))


def _stand_in_batch(first, n):
    """What a rank computes for samples [first, first + n): a function of the GLOBAL sample index only, as the
    pipeline's x_T / conditioning are."""
    pipeline = importlib.import_module("stable-diffusion_amd.pipeline")
    return torch.tanh(pipeline.initial_latents(n, 4, 4, device="cpu", first_sample=first))[:, :3]


def _list_worker(rank, world, port, total, batch, q):
    import sys
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pipeline = importlib.import_module("stable-diffusion_amd.pipeline")
    calls = []

    def run_batch(first, n):
        calls.append((first, n))
        return _stand_in_batch(first, n)
    out = pipeline.run_sample_list(total, batch, run_batch)
    q.put((rank, calls, out))
    dist.barrier()
    dist.destroy_process_group()


def _run_list(world, total, batch):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_list_worker, args=(r, world, port, total, batch, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return got


def test_fixed_sample_list_is_independent_of_world_size():
    """The fixed-list mode of bench.py (64 samples in batches of 8 over G ranks; here the same driver on small
    stand-in images): every rank receives all samples in global order, whatever G is; batches follow the
    reference's chunking of its prompt list (txt2img.py:258-303), the last one of a block may be short."""
    pipeline = importlib.import_module("stable-diffusion_amd.pipeline")
    total, batch = 64, 8
    ref = pipeline.run_sample_list(total, batch, _stand_in_batch, rank=0, world=1)
    assert torch.equal(ref, _stand_in_batch(0, total))
    for world in (2, 4):
        for rank, calls, out in _run_list(world, total, batch):
            lo, hi = pipeline.shard(total, rank, world)
            assert calls == [(s, 8) for s in range(lo, hi, 8)]
            assert torch.equal(out, ref), f"world {world} rank {rank}"


def test_ragged_sample_lists():
    """Totals that do not divide: short last batch, short last block, and a rank that owns nothing."""
    pipeline = importlib.import_module("stable-diffusion_amd.pipeline")
    for world, total, batch in ((2, 7, 2), (4, 5, 2)):
        ref = _stand_in_batch(0, total)
        seen = []
        for rank, calls, out in _run_list(world, total, batch):
            assert torch.equal(out, ref), (world, total, rank)
            seen += [i for first, n in calls for i in range(first, first + n)]
        assert sorted(seen) == list(range(total))
