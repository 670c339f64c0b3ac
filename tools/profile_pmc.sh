#!/bin/bash
# instruction / stall counters of the mid-size evaluation kernels: tools/profile_pmc.sh <outdir> "<counters>" [exp_mid args]
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-prof_pmc}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $2 --kernel-trace --output-format csv -d $OUT/pmc -- python3 $R/tools/exp_mid.py 12 5 --cache /tmp/wl24.pkl --once ${@:3} > $OUT/log.txt 2>&1
python3 - <<PY
import csv,glob,collections
f=glob.glob("$OUT/pmc/*/*counter_collection.csv")
if not f: print(open("$OUT/log.txt").read()[-1500:]); raise SystemExit
acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in csv.DictReader(open(f[0])):
    k=r["Kernel_Name"][:34]
    acc[k][r["Counter_Name"]]+=float(r["Counter_Value"])
for k,v in acc.items():
    print(k, {c: f"{x:.3g}" for c,x in v.items()})
PY
