#!/bin/bash
# instruction / stall counters of the headline kernel: tools/profile_pmc_bench.sh <outdir> "<counters>"
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-prof_pmcb}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $OUT/pmc -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu --no-extra --no-roofline > $OUT/log.txt 2>&1
python3 - <<PY
import csv,glob,collections
f=glob.glob("$OUT/pmc/*/*counter_collection.csv")
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for r in csv.DictReader(open(f[0])):
    k=r["Kernel_Name"][:40]
    acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[(k,r["Counter_Name"])]+=1
for k,v in acc.items():
    if "ovqe" in k: print(k, {c: f"{x:.4g} ({n[(k,c)]} launches)" for c,x in v.items()})
PY
