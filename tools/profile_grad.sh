#!/bin/bash
# kernel stats of the adjoint gradient at 24 qubits (tools/exp_grad.py 12 5 5) under rocprofv3
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_grad
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/tools/exp_grad.py 12 5 5 > $OUT/log.txt 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("$OUT/trace/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    print(r["Name"][:50], r["Calls"], float(r["TotalDurationNs"])/1e6, "ms", float(r["AverageNs"])/1e3, "us")
PY
