#!/bin/bash
# rocprofv3 --kernel-trace --stats of an arbitrary tool script: tools/profile_any.sh <tag> <script.py> [args]; prints the top kernels
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1; shift
OUT=$R/gpurun_out/$tag
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/"$@" > $OUT/run.log 2>&1
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
rm -rf $OUT/trace
python3 - $OUT/kernel_stats.csv <<'PY'
import csv, sys
for i, r in enumerate(csv.DictReader(open(sys.argv[1]))):
    if i < 14: print(f"{r['Name'][:70]:70s} calls {r['Calls']:>6s} total_ms {float(r['TotalDurationNs'])/1e6:9.3f} avg_us {float(r['AverageNs'])/1e3:10.2f} {r['Percentage']}%")
PY
grep -v "^/opt" $OUT/run.log | tail -8
