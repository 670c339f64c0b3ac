#!/usr/bin/env python3
"""Which kernels does which run launch?  Reads every *kernel_stats.csv under the given rocprofv3 output directories (one directory per
run; child processes write their own files) and prints / writes {run: {kernel base name: calls}} next to the list of __global__
kernels in openvqe_amd/csrc/*.hpp.  usage: kernel_reach.py out.json name1=dir1 [name2=dir2 ...]"""
import csv, glob, json, os, re, sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
defined = set()
for f in glob.glob(os.path.join(root, "openvqe_amd", "csrc", "*.hpp")):
    for m in re.finditer(r"__global__[^;{]*?void\s+(k_[a-z0-9_]+)", open(f).read(), re.S):
        defined.add(m.group(1))
out = {"defined": sorted(defined), "runs": {}}
for arg in sys.argv[2:]:
    name, d = arg.split("=", 1)
    acc = {}
    files = glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True) if os.path.isdir(d) else [d]
    for path in files:
        for r in csv.DictReader(open(path)):
            m = re.search(r"\b(k_[a-z0-9_]+)", r["Name"])
            if m:
                acc[m.group(1)] = acc.get(m.group(1), 0) + int(r["Calls"])
    out["runs"][name] = dict(sorted(acc.items()))
    print(name, "files", len(files), "kernels", len(acc), "never launched:", sorted(defined - set(acc)))
json.dump(out, open(sys.argv[1], "w"), indent=1)
