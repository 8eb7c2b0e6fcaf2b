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

stats = glob.glob(os.path.join(ROOT, "gpurun_out", "prof_bench", "*", "*kernel_stats.csv"))
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
    for r in keep:
        if DOM in r["Name"] and "DF16_" in r["Name"]:
            print(f"dominant kernel: calls {r['Calls']}, average {float(r['AverageNs']) / 1e3:.2f} us, {r['Percentage']} %")
    print("wrote", out)


def per_launch(dirname, counter):
    vals = []
    for f in glob.glob(os.path.join(ROOT, "gpurun_out", dirname, "*", "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if DOM in r["Kernel_Name"] and r["Counter_Name"] == counter:
                vals.append(float(r["Counter_Value"]))
    return vals


fetch, write = per_launch("pmc_fetch", "FETCH_SIZE"), per_launch("pmc_write", "WRITE_SIZE")
if fetch and write:
    # rocprofv3 reports both in KiB; on gfx950 FETCH_SIZE counts 128-byte requests as 64 bytes: doubled
    fetch_b = 2.0 * 1024.0 * sum(fetch) / len(fetch)
    write_b = 1024.0 * sum(write) / len(write)
    out = os.path.join(ROOT, "profiles", f"{tag}_pmc_traffic.json")
    json.dump({"kernel": "conv_gemm_pp_kernel<half, PPShape<10,2,2>>", "launches_fetch_pass": len(fetch),
               "launches_write_pass": len(write), "fetch_bytes_per_launch": fetch_b, "write_bytes_per_launch": write_b,
               "hbm_bytes_per_launch": fetch_b + write_b,
               "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over "
                         "bench.py --no-c2 --no-cpu-baseline --no-roofline --no-graph --steps 1 --warmup 0 "
                         "--ddim-steps 4; KiB -> bytes; FETCH_SIZE doubled (gfx950 tallies 128-B requests as 64 B)"},
              open(out, "w"), indent=1)
    print("wrote", out, f"fetch {fetch_b / 1e6:.1f} MB + write {write_b / 1e6:.1f} MB per launch")
