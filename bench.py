#!/usr/bin/env python3
"""Headline benchmark: images/sec for 512x512, 50-step DDIM, classifier-free guidance, on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch: batch-8 txt2img (UNet batch 16 with CFG) = 50 x
[UNet forward + fused CFG/DDIM update] + VAE decode + clamp, preceded by the CLIP text transformer on the
prompts' token ids; fp16 compute, synthetic SD-v1-shaped weights, synthetic token ids standing in for the
tokenizer's output (--no-clip: context tensors instead), inputs resident in HBM.  With N > 1 every
rank runs its own batch (weak scaling; samples are independent trajectories) and the decoded images
are all-gathered over RCCL inside the timed region.  Rank 0 prints one JSON line.

`--gpus N` with N > 1 outside a torchrun launch starts the N ranks itself: the parent process, before it
touches the GPU, runs `python -m torch.distributed.run --nproc-per-node N bench.py <same flags>` as a child,
relays the child's JSON line and exits with its code.  Under torchrun, `--gpus` must equal WORLD_SIZE.

`--total-images T` switches to the fixed-list mode (strong scaling): the T samples of the global list are
dealt to the ranks in contiguous blocks, each rank walks its block in batches of `--batch`, and a "step" is
the whole list including the all-gather; value = T / (max over ranks of the step time).

Extra objects on that line:
  roofline      dominant kernel (implicit-GEMM conv/linear on MFMA): algorithmic FLOPs of its launches /
                their HIP-event time, measured in this process on one extra eager pass of the same step
                (events cannot bracket kernels inside a graph replay), against the 2.5 PFLOP/s dense
                fp16 MFMA peak.
  cpu_baseline  the CPU oracle (fp32 torch restatement of the reference) timed on this host's cores on a
                bounded sample (UNet batch-2 forwards + one VAE decode, extrapolated to 50 steps).
  unet_it_per_s_batch2  the second half of the headline metric: single UNet forward, latent 64x64,
                batch 2, graph replay.
"""
import argparse
import importlib
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_PEAK_TFLOPS = {"fp16": 2500.0, "bf16": 2500.0, "fp32": 157.3}
UNET_GFLOP_PER_SAMPLE = 803.3        # one sample-forward at latent 64x64 (algorithmic, 2*MACs)
VAE_GFLOP_PER_IMAGE = 2514.5
# Work the sampler's CFG-pair shortcut does not execute, per pair and UNet call at latent 64x64: the two halves
# of x_in = cat([x] * 2) are identical up to the first cross-attention, so the first ResBlock (two 3x3 convs
# 320 -> 320 on 4096 pixels), proj_in, the fused q/k/v projection, the 8-head d=40 self-attention over 4096
# tokens, its output projection and the cross-attention query projection run for one of the two samples.
_PX, _D = 4096, 320
CFG_PAIR_SKIPPED_GFLOP = 2 * (2 * 9 * _PX * _D * _D + 6 * _PX * _D * _D + 2 * _PX * _PX * _D) / 1e9


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=8, help="images per GPU per step (per batch with --total-images)")
    ap.add_argument("--total-images", type=int, default=0, help="fixed list of this many samples sharded over "
                    "the ranks in batches of --batch (strong scaling); 0 = one batch per rank (weak scaling)")
    ap.add_argument("--ddim-steps", type=int, default=50)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--dtype", default="fp16", choices=["fp16", "bf16", "fp32"])
    ap.add_argument("--scale", type=float, default=7.5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-clip", action="store_true", help="feed (B,77,768) context tensors instead of running "
                    "the text transformer on token ids inside the step")
    ap.add_argument("--no-parity", action="store_true", help="skip the one extra batch that is held against the "
                    "reference's 50-step image after the timed region")
    ap.add_argument("--fp32-cost", action="store_true", help="also time the fp32 compute mode (UNet batch 2, and a "
                    "batch at 1 and 5 DDIM steps extrapolated to the full count)")
    ap.add_argument("--no-c2", action="store_true", help="skip the batch-2 UNet timing (keeps the kernel "
                    "population of a profiled run identical to the roofline pass)")
    return ap.parse_args()


def prof_read(lib, name):
    import ctypes
    ms, n, fl, by = ctypes.c_double(), ctypes.c_int64(), ctypes.c_double(), ctypes.c_double()
    rc = lib.sdmi_prof_get(name.encode(), ctypes.byref(ms), ctypes.byref(n), ctypes.byref(fl), ctypes.byref(by))
    assert rc == 0, lib.sdmi_last_error()
    return {"ms": ms.value, "launches": n.value, "flops": fl.value, "bytes": by.value}


DOMINANT_KIND = "k1"          # label suffix of the 256x320 large-tile GEMM launches in the profiler dump
DOMINANT_NAME = "conv_gemm_pp_kernel<{elem}, PPShape<10,2,2>> (256x320 implicit-GEMM conv3x3/conv1x1/linear)"


def prof_dump_by_label(lib):
    """Per-launch profiler records -> list of (class, label, us, flops, algorithmic bytes)."""
    import tempfile
    _lib = importlib.import_module("stable-diffusion_amd._lib")
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "launches.csv")
        _lib.check(lib.sdmi_prof_dump(path.encode()), "prof_dump")
        rows = []
        with open(path) as f:
            for i, line in enumerate(f):
                if i == 0:
                    continue
                cls, label, us, fl, by = line.rstrip("\n").split(",")
                rows.append((cls, label, float(us), float(fl), float(by)))
    return rows


def pmc_traffic():
    """(HBM bytes per launch of the dominant kernel, file it came from) from the newest committed rocprofv3 --pmc
    summary (tools/profile_round.sh: counters need passes of their own, so this run cannot measure them)."""
    import glob
    import re
    def round_of(path):
        m = re.match(r"r(\d+)_", os.path.basename(path))
        return int(m.group(1)) if m else -1
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")), key=round_of, reverse=True):
        try:
            with open(path) as f:
                return json.load(f).get("hbm_bytes_per_launch"), os.path.relpath(path, ROOT)
        except (OSError, ValueError):
            continue
    return None, None


def pmc_traffic_classes():
    """Per shape class of the dominant kernel ("conv3x3", "pointwise"): the entries of the same committed summary."""
    path = pmc_traffic()[1]
    if not path:
        return {}
    try:
        with open(os.path.join(ROOT, path)) as f:
            return json.load(f).get("classes") or {}
    except (OSError, ValueError):
        return {}


def cpu_baseline(ddim_steps, size, with_clip=True):
    """Oracle on host cores: 2 UNet batch-2 forwards + 1 VAE decode (+ the text transformer on the empty and one
    prompt), extrapolated to a full image."""
    from oracle import unet_ref, vae_ref, clip_ref
    S = importlib.import_module("stable-diffusion_amd.structure")
    synth = importlib.import_module("stable-diffusion_amd.synth")
    cores = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "0") or 0) or (os.cpu_count() or 1))
    torch.set_num_threads(cores)
    h = size // 8
    st = S.unet_structure(**S.sd_v1_unet_config(4))
    sd = synth.synth_state_dict(S.unet_param_shapes(st), 0)
    x = synth.normal(42, "x_T", (2, 4, h, h))
    ctx = synth.normal(1, "cond_ctx", (2, 77, 768))
    t = torch.full((2,), 981, dtype=torch.long)
    unet_ref.unet_forward(sd, st, x[:, :, :16, :16], t, ctx)          # warm the thread pool
    n_unet = 2
    t0 = time.perf_counter()
    for _ in range(n_unet):
        unet_ref.unet_forward(sd, st, x, t, ctx)
    t_unet = (time.perf_counter() - t0) / n_unet
    del sd
    vst = S.vae_decoder_structure(S.sd_v1_ddconfig(), 4)
    vsd = synth.synth_state_dict(S.vae_decoder_param_shapes(vst), 0)
    t0 = time.perf_counter()
    vae_ref.decode_first_stage(vsd, vst, x[:1])
    t_vae = time.perf_counter() - t0
    t_clip = 0.0
    if with_clip:
        del vsd
        clip = importlib.import_module("stable-diffusion_amd.clip")
        pipeline = importlib.import_module("stable-diffusion_amd.pipeline")
        csd = synth.synth_state_dict(clip.clip_text_param_shapes(), 0)
        ids = torch.cat(pipeline.synthetic_prompt_tokens(1, "cpu"))
        t0 = time.perf_counter()
        clip_ref.clip_text_forward(csd, ids)
        t_clip = time.perf_counter() - t0
    per_image = ddim_steps * t_unet + t_vae + t_clip
    return {"value": 1.0 / per_image, "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": f"{n_unet} UNet batch-2 forwards ({t_unet:.2f} s each) + 1 VAE decode ({t_vae:.2f} s) "
                      f"+ text transformer on 2 token rows ({t_clip:.2f} s) at "
                      f"{size}x{size}, fp32 torch-CPU oracle, extrapolated to {ddim_steps} steps per image",
            "unet_it_per_s_batch2": 1.0 / t_unet}


_T0 = time.perf_counter()


def log(rank, msg):
    """Progress on stderr (stdout carries the one JSON line)."""
    print(f"[bench rank {rank} +{time.perf_counter() - _T0:6.1f}s] {msg}", file=sys.stderr, flush=True)


def visible_gpu_count():
    """GPUs this process could open, counted without any HIP / torch.cuda call: the KFD topology nodes that have
    SIMDs, capped by the *_VISIBLE_DEVICES lists.  An upper bound (a container may see nodes it cannot open); a rank
    that finds no device of its own still fails on its own and its code is relayed."""
    import glob
    n = 0
    for path in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
        try:
            with open(path) as f:
                for line in f:
                    if line.startswith("simd_count ") and int(line.split()[1]) > 0:
                        n += 1
        except (OSError, ValueError):
            continue
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        if var in os.environ:
            n = min(n, len([e for e in os.environ[var].split(",") if e.strip()]))
    return n


def spawn_ranks(args):
    """Parent of a multi-rank run that was not started by torchrun: start the ranks as a child process, pass the
    rank-0 JSON line through and return the child's code.  This process makes no torch.cuda / HIP call at all: devices
    are counted from sysfs and the environment."""
    import socket
    import subprocess
    have = visible_gpu_count()
    if have < args.gpus and os.environ.get("SDMI_BENCH_SHARE_DEVICE") != "1":
        print(f"bench.py: --gpus {args.gpus} but only {have} device(s) are visible", file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for line in proc.stdout:
        # stdout carries exactly the JSON line; anything else a rank or the launcher printed goes to stderr
        (sys.stdout if line.startswith("{") else sys.stderr).write(line)
        sys.stdout.flush()
    return proc.wait()


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        print(f"bench.py: --gpus {args.gpus} does not match WORLD_SIZE {world}", file=sys.stderr)
        sys.exit(2)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch.distributed as dist
    # rehearsal knobs for a box with fewer GPUs than ranks: SDMI_BENCH_BACKEND=gloo (no RCCL between ranks that
    # share a device) and SDMI_BENCH_SHARE_DEVICE=1 (every rank on cuda:0); the driver's runs use neither
    backend = os.environ.get("SDMI_BENCH_BACKEND", "nccl")
    if os.environ.get("SDMI_BENCH_SHARE_DEVICE") == "1":
        local_rank = 0
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", torch.cuda.current_device())

    pipeline = importlib.import_module("stable-diffusion_amd.pipeline")
    sdmi = importlib.import_module("stable-diffusion_amd")
    _lib = importlib.import_module("stable-diffusion_amd._lib")
    lib = _lib.lib()

    log(rank, f"world {world}, device {dev}, building the model")
    model = pipeline.build_synthetic_model(4, "crossattn", args.dtype, device=dev, text_encoder=not args.no_clip)
    log(rank, "model built")
    unet = model.model.diffusion_model
    if args.no_graph:
        unet.set_graph_mode(False)
    sampler = sdmi.DDIMSampler(model)
    B, h = args.batch, args.size // 8
    total = args.total_images if args.total_images > 0 else world * B
    lo, hi = pipeline.shard(total, rank, world)        # this rank's block of the global sample list

    # inputs of every batch of the block, derived from the global sample index and resident in HBM
    def batch_inputs(first, n):
        if args.no_clip:
            uc, c = pipeline.synthetic_conditioning(n, dev, first_sample=first)
        else:
            uc, c = pipeline.synthetic_prompt_tokens(n, dev, first_sample=first)
        return pipeline.initial_latents(n, h, h, dev, first_sample=first), uc, c
    inputs = {first: batch_inputs(first, n) for first, n in pipeline.sample_batches(lo, hi, B)}

    last = {}

    def run_batch(first, n):
        # decoded images leave the step as the uint8 HWC array the reference's script saves (txt2img.py:307-317):
        # that is what the ranks gather
        x_T, uc, c = inputs[first]
        img, last["z"] = pipeline.txt2img(model, x_T, uc, c, steps=args.ddim_steps, scale=args.scale, sampler=sampler,
                                          as_uint8=True)
        return img

    def local_step():
        return [run_batch(first, n) for first, n in pipeline.sample_batches(lo, hi, B)]

    def step():
        return pipeline.gather_sharded(local_step(), total, rank, world)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    log(rank, f"{args.warmup} warmup step(s) done, timing {args.steps}")
    # SDMI_BENCH_LABELS=path (profiling runs, eager mode): the per-launch shape labels of the timed steps, in launch
    # order, so that a counter pass over the same command can be split by shape class
    labels_path = os.environ.get("SDMI_BENCH_LABELS") if args.no_graph and rank == 0 else None
    if labels_path:
        lib.sdmi_prof_reset()
        lib.sdmi_prof_enable(1)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    fence()
    elapsed = time.perf_counter() - t0
    log(rank, f"timed region {elapsed:.3f} s")
    if labels_path:
        lib.sdmi_prof_enable(0)
        _lib.check(lib.sdmi_prof_dump(labels_path.encode()), "prof_dump")
        lib.sdmi_prof_reset()
    if world > 1:
        tt = torch.tensor([elapsed], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    assert out.shape[0] == total and out.dtype == torch.uint8 and bool(torch.isfinite(last["z"]).all())
    assert 0 < int(out.max()) and int(out.min()) < 255, "degenerate images"
    images = total * args.steps
    value = images / elapsed

    # the gather on its own (it is also inside the timed region): same payload, 5 repetitions, max over ranks
    per_rank = (total + world - 1) // world
    collective = {"op": "all_gather_into_tensor of the uint8 images", "backend": None, "ranks": 1,
                  "bytes_per_rank": int(per_rank * out[0].numel() * out.element_size()), "ms": 0.0}
    if world > 1:
        parts = local_step()
        fence()
        t0 = time.perf_counter()
        for _ in range(5):
            pipeline.gather_sharded(parts, total, rank, world)
        fence()
        tg = torch.tensor([(time.perf_counter() - t0) / 5], device=dev if backend == "nccl" else "cpu",
                          dtype=torch.float64)
        dist.all_reduce(tg, op=dist.ReduceOp.MAX)
        collective.update(backend=dist.get_backend(), ranks=dist.get_world_size(), ms=1e3 * float(tg.item()))

    result = {
        "metric": "images/sec (node) 512x512 50-step DDIM CFG; UNet it/s at batch 2",
        "value": value, "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
        "scaling": "strong" if args.total_images > 0 else "weak",
        "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"txt2img {args.size}x{args.size}, {args.ddim_steps} DDIM steps, CFG {args.scale}, "
                               + (f"fixed list of {total} samples in batches of {B} (UNet batch {2 * B}), "
                                  if args.total_images > 0 else f"batch {B}/GPU (UNet batch {2 * B}), ")
                               + ("context tensors in place of CLIP" if args.no_clip else
                                  "CLIP text transformer on synthetic token ids") +
                               " + UNet loop + fused CFG/DDIM update + VAE decode + clamp; SD-v1-shaped random weights",
                   "global_batch": total, "latent": [4, h, h], "parallelism": f"replicas x{world} (sample shard)",
                   "graph_replay": not args.no_graph},
        "collective": collective,
    }
    if rank == 0 and not args.no_graph:
        result["unet_launches_per_forward"] = {f"batch{2 * B}_sampler_mode": unet.graph_launches()}

    if rank == 0 and not args.no_c2:
        # ---- C2: single UNet forward, batch 2 -------------------------------------------------------
        x2 = pipeline.initial_latents(1, h, h, dev).repeat(2, 1, 1, 1).contiguous()
        uc2, c2 = pipeline.synthetic_conditioning(1, dev)
        ctx2 = torch.cat([uc2, c2])
        t2 = torch.full((2,), 981, dtype=torch.int64, device=dev)
        o2 = torch.empty_like(x2)
        for _ in range(3):
            unet(x2, t2, context=ctx2, out=o2)
        torch.cuda.synchronize()
        n2 = 30
        t0 = time.perf_counter()
        for _ in range(n2):
            unet(x2, t2, context=ctx2, out=o2)
        torch.cuda.synchronize()
        dt2 = (time.perf_counter() - t0) / n2
        result["unet_it_per_s_batch2"] = 1.0 / dt2
        result["unet_batch2_ms"] = 1e3 * dt2
        result["unet_batch2_tflops"] = 2 * UNET_GFLOP_PER_SAMPLE / dt2 / 1e3
        if not args.no_graph:
            result["unet_launches_per_forward"]["batch2"] = unet.graph_launches()
        # whole-step algorithmic rate
        if args.size == 512:
            # whole-step rate on the work actually executed (the CFG-pair shortcut's skipped layers are not counted)
            per_image = (2 * UNET_GFLOP_PER_SAMPLE - CFG_PAIR_SKIPPED_GFLOP) * args.ddim_steps + VAE_GFLOP_PER_IMAGE
            result["step_tflops"] = total * per_image / 1e3 / (elapsed / args.steps)
            result["step_tflop_executed"] = total * per_image / 1e3

    if rank == 0 and not args.no_parity and args.size == 512 and args.ddim_steps == 50 and args.scale == 7.5 \
            and args.dtype != "fp32":
        # ---- the timed sequence against the reference: one more batch through exactly the path the timed region
        # ran (same batch, 50 steps, graph replay, sampler promises), in which slots 0 and 5 carry the x_T and the
        # conditioning of the reference's own 50-step trajectory (tests/golden/g7_c3_traj.npz, made by running the
        # unmodified reference on these weights): PSNR of slot 0 against the reference's image, and slots 0 and 5
        # must come out identical.  Outside the timed region.
        log(rank, "parity of the timed sequence against the reference's 50-step image")
        import numpy as np
        gz = np.load(os.path.join(ROOT, "tests", "golden", "g7_c3_traj.npz"), allow_pickle=False)
        gt = lambda k: torch.from_numpy(np.asarray(gz[k])).float().to(dev)
        x_T = pipeline.initial_latents(B, h, h, dev).clone()
        uc, c = pipeline.synthetic_conditioning(B, dev)
        uc, c = uc.clone(), c.clone()
        for slot in (0, 5 % B):
            x_T[slot], uc[slot], c[slot] = gt("x_T")[0], gt("uc")[0], gt("c")[0]
        img, _ = pipeline.txt2img(model, x_T, uc, c, steps=50, scale=7.5, sampler=sampler)
        torch.cuda.synchronize()
        ref = gt("image")[0]
        mse = float(((img[0].double() - ref.double()) ** 2).mean())
        result["parity_psnr_db"] = 99.0 if mse == 0 else float(10.0 * np.log10(1.0 / mse))
        result["parity_slots_identical"] = bool(torch.equal(img[0], img[5 % B]))
        result["parity_reference"] = "tests/golden/g7_c3_traj.npz (the reference's fp32 CPU image of the same weights)"
        del img, x_T, uc, c

    if rank == 0 and args.fp32_cost and args.dtype != "fp32":
        # ---- what the drop-in default (fp32 compute, the reference's use_fp16=False) costs on the same box ----
        log(rank, "fp32 mode: UNet batch 2 and a reduced-step batch")
        m32 = pipeline.build_synthetic_model(4, "crossattn", "fp32", device=dev)
        u32 = m32.model.diffusion_model
        x2 = pipeline.initial_latents(1, h, h, dev).repeat(2, 1, 1, 1).contiguous()
        uc2, c2 = pipeline.synthetic_conditioning(1, dev)
        ctx2 = torch.cat([uc2, c2])
        t2 = torch.full((2,), 981, dtype=torch.int64, device=dev)
        o2 = torch.empty_like(x2)
        for _ in range(3):
            u32(x2, t2, context=ctx2, out=o2)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            u32(x2, t2, context=ctx2, out=o2)
        torch.cuda.synchronize()
        result["fp32_unet_batch2_ms"] = 1e3 * (time.perf_counter() - t0) / 5
        S32 = 5            # (1 and 5 both divide the 1000 training steps evenly, which make_ddim_timesteps needs)
        s32 = sdmi.DDIMSampler(m32)
        xT = pipeline.initial_latents(B, h, h, dev)
        uc8, c8 = pipeline.synthetic_conditioning(B, dev)
        def run32(steps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            pipeline.txt2img(m32, xT, uc8, c8, steps=steps, scale=args.scale, sampler=s32)
            torch.cuda.synchronize()
            return time.perf_counter() - t0
        run32(1)
        t1, t3 = run32(1), run32(S32)
        per_step = (t3 - t1) / (S32 - 1)          # one UNet batch-16 call + update
        fixed = t1 - per_step                      # VAE decode + clamp
        result["fp32_images_per_s"] = B / (args.ddim_steps * per_step + fixed)
        result["fp32_note"] = (f"fp32 compute mode (the drop-in default): batch {B}, measured at 1 and {S32} DDIM steps "
                               f"({1e3 * per_step:.1f} ms per step, {1e3 * fixed:.1f} ms decode) and extrapolated to "
                               f"{args.ddim_steps} steps; every step is the same UNet call")
        del m32, u32, s32
        torch.cuda.empty_cache()

    if rank == 0 and not args.no_roofline:
        log(rank, "roofline pass")
        # ---- roofline of the dominant kernel: one extra eager pass of the same step under HIP events ----
        unet.set_graph_mode(False)
        lib.sdmi_prof_reset()
        lib.sdmi_prof_enable(1)
        run_batch(*pipeline.sample_batches(lo, hi, B)[0])      # rank 0 alone, one batch: no collective in this pass
        torch.cuda.synchronize()
        lib.sdmi_prof_enable(0)
        gemm = prof_read(lib, "conv_gemm")
        others = {k: prof_read(lib, k) for k in ("attention", "groupnorm", "layernorm")}
        dom = [r for r in prof_dump_by_label(lib) if r[0] == "conv_gemm" and r[1].endswith(" " + DOMINANT_KIND)]
        lib.sdmi_prof_reset()
        unet.set_graph_mode(not args.no_graph)
        peak = MFMA_PEAK_TFLOPS[args.dtype]
        # an event bracket is a little longer than the kernel inside it: subtract the empty-bracket time
        import ctypes
        cal = ctypes.c_double()
        _lib.check(lib.sdmi_prof_calibrate(200, _lib.stream_ptr(), ctypes.byref(cal)), "prof_calibrate")
        raw_us, dom_fl = sum(r[2] for r in dom), sum(r[3] for r in dom)
        dom_us = max(raw_us - cal.value * len(dom), 1e-9)
        achieved = dom_fl / (dom_us * 1e-6) / 1e12 if dom else 0.0
        def shape_class(rows_):
            us = max(sum(r[2] for r in rows_) - cal.value * len(rows_), 1e-9)
            fl = sum(r[3] for r in rows_)
            tf = fl / (us * 1e-6) / 1e12 if rows_ else 0.0
            return {"achieved": tf, "frac": tf / peak, "launches": len(rows_), "kernel_ms_per_step": us * 1e-3,
                    "avg_launch_us": us / max(1, len(rows_)), "flops_per_launch": fl / max(1, len(rows_)),
                    "algorithmic_bytes_per_launch": sum(r[4] for r in rows_) / max(1, len(rows_))}
        conv3 = shape_class([r for r in dom if " t9 " in r[1]])
        point = shape_class([r for r in dom if " t1 " in r[1]])
        tr = pmc_traffic_classes()
        for cls, name in ((conv3, "conv3x3"), (point, "pointwise")):
            cls["traffic"] = (tr.get(name) or {}).get("hbm_bytes_per_launch")
            cls["sustained_clock_GHz"] = (tr.get(name) or {}).get("shader_clock_GHz")
        result["roofline"] = {
            "conv3x3": conv3, "pointwise": point,
            "bound": "mfma", "kernel": DOMINANT_NAME.format(elem={"fp16": "half", "bf16": "bfloat16"}.get(args.dtype, "float")),
            "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak, "traffic": pmc_traffic()[0],
            "traffic_source": pmc_traffic()[1],
            "launches": len(dom), "avg_launch_us": dom_us / max(1, len(dom)),
            "flops_per_launch": dom_fl / max(1, len(dom)), "kernel_ms_per_step": dom_us * 1e-3,
            "algorithmic_bytes_per_launch": sum(r[4] for r in dom) / max(1, len(dom)),
            "event_bracket_us": {"raw_avg": raw_us / max(1, len(dom)), "empty_bracket": cal.value},
            "all_gemm_kernels": {"achieved": gemm["flops"] / (gemm["ms"] * 1e-3) / 1e12, "launches": gemm["launches"],
                                 "kernel_ms_per_step": gemm["ms"]},
        }
        tot = gemm["ms"] + sum(v["ms"] for v in others.values())
        result["kernel_time_share"] = {"conv_gemm": gemm["ms"] / tot,
                                       **{k: v["ms"] / tot for k, v in others.items()}}
        att = others["attention"]
        result["attention_tflops"] = att["flops"] / (att["ms"] * 1e-3) / 1e12 if att["ms"] > 0 else None
        for k in ("groupnorm", "layernorm"):
            v = others[k]
            result[k + "_GBps"] = v["bytes"] / (v["ms"] * 1e-3) / 1e9 if v["ms"] > 0 else None

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        log(rank, "CPU baseline (about a minute of host work)")
        result["cpu_baseline"] = cpu_baseline(args.ddim_steps, args.size, with_clip=not args.no_clip)

    log(rank, "done")
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()
