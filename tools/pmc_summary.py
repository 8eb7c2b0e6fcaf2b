#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc output: pmc_summary.py <dir> <kernel-name-substring>
Prints, per counter, the mean value over the matching dispatches."""
import collections, csv, glob, sys
d, pat = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
dur = []
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    with open(f, newline="") as fh:
        for row in csv.DictReader(fh):
            if pat in row["Kernel_Name"]:
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
                dur.append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
for k in sorted(acc):
    v = acc[k]
    print(f"{k:32s} n={len(v):3d} mean={sum(v) / len(v):16.1f}")
if dur:
    print(f"dispatch duration mean {sum(dur) / len(dur):.1f} us (profiled)")
