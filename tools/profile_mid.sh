#!/bin/bash
# kernel-time vs wall-time of a mid-size evaluation: tools/profile_mid.sh <m> <o> <outdir>
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${3:-prof_mid}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/tools/exp_mid.py $1 $2 > $OUT/trace.log 2>&1
grep prepare $OUT/trace.log
python3 - <<PY
import csv,glob
f=glob.glob("$OUT/trace/*/*kernel_stats.csv")[0]
tot=0
for r in csv.DictReader(open(f)):
    print(r["Name"][:50], r["Calls"], float(r["TotalDurationNs"])/1e6, "ms", float(r["AverageNs"])/1e3, "us")
    tot+=float(r["TotalDurationNs"])
print("total kernel ms", tot/1e6)
PY
