#!/usr/bin/env python3
"""Condense the raw rocprofv3 output of tools/profile_round.sh (gpurun_out/) into profiles/:
   profiles/<tag>_bench_kernel_stats.csv   per-kernel totals / averages of the bench command
   profiles/<tag>_pmc_traffic.json         HBM bytes per launch of the dominant GEMM kernel
usage: profile_summary.py <tag>   (e.g. r01)"""
import csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
DOM = "PPShapeILi10ELi2ELi2E"          # mangled fragment of conv_gemm_pp_kernel<half, PPShape<10,2,2>>
os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)


def newest(pattern):
    """The most recently written file matching `pattern` (gpurun_out keeps the output of earlier runs, named by
    process id, next to the latest one)."""
    files = glob.glob(pattern)
    return [max(files, key=os.path.getmtime)] if files else []


stats = newest(os.path.join(ROOT, "gpurun_out", "prof_bench", "*", "*kernel_stats.csv"))
if stats:
    rows = list(csv.DictReader(open(stats[0])))
    keep = [r for r in rows if "at::native" not in r["Name"]]
    out = os.path.join(ROOT, "profiles", f"{tag}_bench_kernel_stats.csv")
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in keep:
            w.writerow([r["Name"][:160], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"],
                        r["MaxNs"]])
    # the 256x320 shape exists in two instantiations (K order); bench.py's roofline covers both
    dom = [r for r in keep if DOM in r["Name"] and "DF16_" in r["Name"]]
    for r in dom:
        print(f"dominant kernel: calls {r['Calls']}, average {float(r['AverageNs']) / 1e3:.2f} us, {r['Percentage']} %")
    if dom:
        calls = sum(int(r["Calls"]) for r in dom)
        tot = sum(float(r["TotalDurationNs"]) for r in dom)
        json.dump({"kernel": "conv_gemm_pp_kernel<half, PPShape<10,2,2>, TI = false | true>", "calls": calls,
                   "average_us": tot / calls / 1e3, "share_of_gpu_time_percent": sum(float(r["Percentage"]) for r in dom),
                   "instantiations": [{"name": r["Name"][:150], "calls": int(r["Calls"]),
                                       "average_us": float(r["AverageNs"]) / 1e3} for r in dom]},
                  open(os.path.join(ROOT, "profiles", f"{tag}_dominant_kernel.json"), "w"), indent=1)
        print(f"dominant kernel, both K orders: calls {calls}, average {tot / calls / 1e3:.2f} us")
    print("wrote", out)


def per_launch(dirname, counter):
    """Values of `counter` for the dominant kernel's dispatches, in dispatch order."""
    vals = []
    for f in newest(os.path.join(ROOT, "gpurun_out", dirname, "*", "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if DOM in r["Kernel_Name"] and r["Counter_Name"] == counter:
                vals.append((int(r["Dispatch_Id"]), float(r["Counter_Value"]),
                             (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
    vals.sort()
    return vals


def dominant_labels():
    """Shape labels of the dominant kernel's launches in launch order (bench.py with SDMI_BENCH_LABELS)."""
    path = os.path.join(ROOT, "gpurun_out", "pmc_labels.csv")
    if not os.path.exists(path):
        return []
    out = []
    with open(path) as f:
        next(f)
        for line in f:
            cls, label = line.split(",")[:2]
            if cls == "conv_gemm" and label.endswith(" k1"):
                out.append("conv3x3" if " t9 " in label else "pointwise")
    return out


fetch, write = per_launch("pmc_fetch", "FETCH_SIZE"), per_launch("pmc_write", "WRITE_SIZE")
if fetch and write:
    # rocprofv3 reports both in KiB; on gfx950 FETCH_SIZE counts 128-byte requests as 64 bytes: doubled
    mean_v = lambda rows: sum(r[1] for r in rows) / len(rows)
    fetch_b = 2.0 * 1024.0 * mean_v(fetch)
    write_b = 1024.0 * mean_v(write)
    doc = {"kernel": "conv_gemm_pp_kernel<half, PPShape<10,2,2>>", "launches_fetch_pass": len(fetch),
           "launches_write_pass": len(write), "fetch_bytes_per_launch": fetch_b, "write_bytes_per_launch": write_b,
           "hbm_bytes_per_launch": fetch_b + write_b,
           "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over "
                     "bench.py --no-c2 --no-cpu-baseline --no-roofline --no-graph --steps 1 --warmup 0 "
                     "--ddim-steps 4; KiB -> bytes; FETCH_SIZE doubled (gfx950 tallies 128-B requests as 64 B); "
                     "classes: the i-th dispatch of the kernel is the i-th launch label of the same command "
                     "(taps = 9: conv3x3, taps = 1: pointwise)"}
    labels = dominant_labels()
    if len(labels) == len(fetch) == len(write):
        doc["classes"] = {}
        for name in ("conv3x3", "pointwise"):
            fr = [r for r, l in zip(fetch, labels) if l == name]
            wr = [r for r, l in zip(write, labels) if l == name]
            if not fr:
                continue
            fb, wb = 2.0 * 1024.0 * mean_v(fr), 1024.0 * mean_v(wr)
            doc["classes"][name] = {"launches": len(fr), "fetch_bytes_per_launch": fb, "write_bytes_per_launch": wb,
                                    "hbm_bytes_per_launch": fb + wb,
                                    "avg_duration_us": sum(r[2] for r in fr + wr) / (len(fr) + len(wr))}
    else:
        print(f"labels ({len(labels)}) do not line up with the dispatches ({len(fetch)}, {len(write)}): no classes")
    # sustained clock per class from the SQ pass, joined the same way
    sqb = per_launch("pmc_sq", "SQ_BUSY_CYCLES")
    if "classes" in doc and len(sqb) == len(labels):
        for name, c in doc["classes"].items():
            rows = [r for r, l in zip(sqb, labels) if l == name]
            c["shader_clock_GHz"] = sum(r[1] for r in rows) / 32.0 / sum(r[2] for r in rows) / 1e3
    out = os.path.join(ROOT, "profiles", f"{tag}_pmc_traffic.json")
    json.dump(doc, open(out, "w"), indent=1)
    print("wrote", out, f"fetch {fetch_b / 1e6:.1f} MB + write {write_b / 1e6:.1f} MB per launch")
    for name, c in doc.get("classes", {}).items():
        print(f"  {name}: {c['launches']} launches, {c['hbm_bytes_per_launch'] / 1e6:.1f} MB per launch, "
              f"{c['avg_duration_us']:.1f} us, clock {c.get('shader_clock_GHz', 0):.2f} GHz")


# ---- per kernel class: SQ issue/busy counters and HBM traffic of the normalisation kernels -----------------
def classify(k):
    if DOM in k:
        return "conv_gemm 256x320"
    if "conv_gemm_pp" in k:
        return "conv_gemm other large tiles"
    if "conv_gemm_kernel" in k:
        return "conv_gemm 128x128"
    if "attn_flash" in k:
        return "attn_flash"
    if "xattn_resident" in k:
        return "xattn_resident"
    if "cross_attn" in k:
        return "cross_attn"
    if "gn_partial" in k:
        return "gn_partial"
    if "gn_apply" in k:
        return "gn_apply"
    if "gn_group" in k:
        return "gn_group (one launch)"
    if "layernorm" in k:
        return "layernorm"
    return None


def collect(dirname):
    """class -> counter -> list of per-dispatch values (+ 'dur_us' once per dispatch)."""
    out = {}
    for f in newest(os.path.join(ROOT, "gpurun_out", dirname, "*", "*counter_collection.csv")):
        seen = set()
        for r in csv.DictReader(open(f)):
            c = classify(r["Kernel_Name"])
            if c is None:
                continue
            d = out.setdefault(c, {})
            d.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
            if r["Dispatch_Id"] not in seen:
                seen.add(r["Dispatch_Id"])
                d.setdefault("dur_us", []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    return out


mean = lambda v: sum(v) / len(v)
sq = collect("pmc_sq")
nf, nw = collect("pmc_norm_fetch"), collect("pmc_norm_write")
summary = {"units": "per dispatch means over a 4-DDIM-step eager run of bench.py (UNet batch 16 + VAE decode); "
                    "SQ_VALU_MFMA_BUSY_CYCLES in cycles summed over the 1024 SIMDs (= 16 x the number of "
                    "16x16x32 MFMAs, 32 x for 32x32x16); SQ_WAVE_CYCLES / SQ_ACTIVE_INST_* / SQ_WAIT_ANY in "
                    "quad-cycles summed over waves; FETCH_SIZE / WRITE_SIZE in KiB, FETCH_SIZE doubled for gfx950",
           "sq": {}, "hbm": {}}
for c, d in sq.items():
    waves_cyc = 4.0 * mean(d["SQ_WAVE_CYCLES"])
    summary["sq"][c] = {
        "dispatches": len(d["dur_us"]), "avg_duration_us": mean(d["dur_us"]),
        "mfma_busy_cycles_per_simd": mean(d["SQ_VALU_MFMA_BUSY_CYCLES"]) / 1024.0,
        "valu_active_cycles_per_simd": 4.0 * mean(d["SQ_ACTIVE_INST_VALU"]) / 1024.0,
        "lds_active_cycles_per_simd": 4.0 * mean(d["SQ_ACTIVE_INST_LDS"]) / 1024.0,
        "wave_cycles_per_simd": waves_cyc / 1024.0,
        "wait_any_share_of_wave_cycles": mean(d["SQ_WAIT_ANY"]) / mean(d["SQ_WAVE_CYCLES"]),
        "mfma_busy_share_of_wave_cycles": mean(d["SQ_VALU_MFMA_BUSY_CYCLES"]) / waves_cyc,
        # SQ_BUSY_CYCLES is summed over the 32 shader engines (8 XCDs x 4): /32 = cycles the kernel kept them busy
        "kernel_cycles": mean(d["SQ_BUSY_CYCLES"]) / 32.0,
        "shader_clock_GHz": mean(d["SQ_BUSY_CYCLES"]) / 32.0 / mean(d["dur_us"]) / 1e3,
        "mfma_pipe_busy_share": (mean(d["SQ_VALU_MFMA_BUSY_CYCLES"]) / 1024.0) / (mean(d["SQ_BUSY_CYCLES"]) / 32.0),
        "valu_issue_share": (4.0 * mean(d["SQ_ACTIVE_INST_VALU"]) / 1024.0) / (mean(d["SQ_BUSY_CYCLES"]) / 32.0),
        "valu_instructions": mean(d["SQ_INSTS_VALU"]),
    }
for c in nf:
    if c not in nw:
        continue
    fb = 2.0 * 1024.0 * mean(nf[c]["FETCH_SIZE"])
    wb = 1024.0 * mean(nw[c]["WRITE_SIZE"])
    dur = mean(nf[c]["dur_us"] + nw[c]["dur_us"])
    summary["hbm"][c] = {"dispatches": len(nf[c]["dur_us"]), "avg_duration_us": dur, "fetch_bytes": fb, "write_bytes": wb,
                         "achieved_GBps": (fb + wb) / dur / 1e3, "peak_GBps": 8000.0}
if summary["sq"] or summary["hbm"]:
    out = os.path.join(ROOT, "profiles", f"{tag}_pmc_summary.json")
    json.dump(summary, open(out, "w"), indent=1)
    print("wrote", out)
    for c, v in summary["sq"].items():
        print(f"  {c:30s} {v['avg_duration_us']:7.1f} us  MFMA pipe busy {100 * v['mfma_pipe_busy_share']:5.1f}% of kernel cycles "
              f"at {v['shader_clock_GHz']:.2f} GHz, VALU issue {100 * v['valu_issue_share']:5.1f}%, waves waiting "
              f"{100 * v['wait_any_share_of_wave_cycles']:5.1f}% of their cycles")
    for c, v in summary["hbm"].items():
        print(f"  {c:30s} {v['avg_duration_us']:7.1f} us  {v['achieved_GBps']:7.1f} GB/s")
