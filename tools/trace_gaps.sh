#!/bin/bash
# where a run's device sits idle: rocprofv3 --kernel-trace of a tool script, then every gap above <us> microseconds between consecutive
# dispatches with the kernels on both sides and the busy time since the previous listed gap;  usage: tools/trace_gaps.sh <name> <us> tools/<script.py> [args]
name=$1; us=$2; shift; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
out=$R/gpurun_out/$name
mkdir -p $out
rocprofv3 --kernel-trace --output-format csv -d /tmp/tg_$name -o t -- python3 $R/"$@" > $out/run.log 2>&1
python3 - /tmp/tg_$name $us > $out/gaps.txt <<'PY'
import csv, glob, sys
rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
t0 = int(rows[0]['Start_Timestamp']); prev = None; busy = 0.0; n = 0
for r in rows:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if prev is not None and (s - int(prev['End_Timestamp'])) / 1e3 > float(sys.argv[2]):
        print(f"at {(s - t0) / 1e6:9.3f} ms: idle {(s - int(prev['End_Timestamp'])) / 1e3:9.1f} us after {n:4d} dispatches busy {busy / 1e3:9.1f} us | {prev['Kernel_Name'][:50]} -> {r['Kernel_Name'][:50]}")
        busy = 0.0; n = 0
    busy += (e - s); n += 1
    prev = r
PY
tail -2 $out/run.log | cut -c1-250; cat $out/gaps.txt
