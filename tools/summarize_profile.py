#!/usr/bin/env python3
"""Collect the rocprofv3 outputs of tools/profile_bench.sh / profile_n24.sh (gpurun_out/<dir>) into the two small
CSVs kept under profiles/<round>/: kernel_stats.csv (the --stats table) and pmc_summary.csv (per kernel and counter:
dispatches, mean / min / max of the raw counter value in KiB as rocprofv3 reports FETCH_SIZE / WRITE_SIZE).
usage: summarize_profile.py gpurun_out/prof profiles/r1e"""
import collections
import csv
import glob
import os
import shutil
import sys

src, dst = sys.argv[1], sys.argv[2]
os.makedirs(dst, exist_ok=True)
stats = glob.glob(os.path.join(src, "trace", "*", "*kernel_stats.csv"))
if stats:
    shutil.copy(stats[0], os.path.join(dst, "kernel_stats.csv"))
acc = collections.OrderedDict()
for sub in ("pmc_fetch", "pmc_write"):
    for path in glob.glob(os.path.join(src, sub, "*", "*counter_collection.csv")):
        for r in csv.DictReader(open(path)):
            acc.setdefault((r["Kernel_Name"], r["Counter_Name"]), []).append(float(r["Counter_Value"]))
with open(os.path.join(dst, "pmc_summary.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "counter", "dispatches", "mean_KiB", "min_KiB", "max_KiB"])
    for (k, c), v in acc.items():
        w.writerow([k, c, len(v), sum(v) / len(v), min(v), max(v)])
print("wrote", dst, "kernels:", len({k for k, _ in acc}))
