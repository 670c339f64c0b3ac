#!/bin/bash
# kernel timeline of the end of a run: rocprofv3 --kernel-trace of a tool script, then start offset / duration / gap to the previous
# kernel of the last N dispatches;  usage: tools/trace_timeline.sh <name> <N> tools/<script.py> [args]
name=$1; n=$2; shift; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
out=$R/gpurun_out/$name
mkdir -p $out
rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$name -o t -- python3 $R/"$@" > $out/run.log 2>&1
python3 - /tmp/tl_$name $n > $out/timeline.txt <<'PY'
import csv, glob, sys
rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
rows = rows[-int(sys.argv[2]):]
t0 = int(rows[0]['Start_Timestamp']); prev_end = None
for r in rows:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = '' if prev_end is None else f"{(s - prev_end) / 1e3:7.2f}"
    print(f"{(s - t0) / 1e3:9.2f} us  dur {(e - s) / 1e3:7.2f} us  gap {gap:>7s} us  grid {r['Grid_Size_X']:>8s} wg {r['Workgroup_Size_X']:>5s}  {r['Kernel_Name'][:80]}")
    prev_end = e
PY
tail -3 $out/run.log; cat $out/timeline.txt
