#!/bin/bash
# HIP API time of a tool script by function: rocprofv3 --hip-trace --stats; usage: tools/profile_api.sh <name> tools/<script.py> [args]
name=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
out=$R/gpurun_out/$name
mkdir -p $out
rocprofv3 --hip-trace --stats --output-format csv -d /tmp/pa_$name -o t -- python3 $R/"$@" > $out/run.log 2>&1
f=$(find /tmp/pa_$name -name '*hip_api_stats.csv' | head -1)
cp $f $out/hip_api_stats.csv
python3 - $f <<'PY'
import csv, sys
for i, r in enumerate(csv.DictReader(open(sys.argv[1]))):
    if i < 16: print(f"{r['Name'][:40]:40s} calls {r['Calls']:>7s} total_ms {float(r['TotalDurationNs'])/1e6:9.2f} avg_us {float(r['AverageNs'])/1e3:10.1f} max_ms {float(r['MaxNs'])/1e6:8.2f}")
PY
grep -v "^/opt\|rocprofv3" $out/run.log | tail -3 | cut -c1-300
