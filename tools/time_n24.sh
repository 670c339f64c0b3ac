#!/bin/bash
# 24-qubit M3 workload: wall time of one evaluation + per-launch durations of the <H> sweeps under rocprofv3
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-t24}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/tools/exp_mid.py 12 5 --cache /tmp/wl24.pkl > /dev/null 2>&1
shift
python3 $R/tools/exp_mid.py 12 5 --cache /tmp/wl24.pkl "$@" 2>&1 | grep prepare | tail -1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/tools/exp_mid.py 12 5 --cache /tmp/wl24.pkl "$@" > /dev/null 2>&1
python3 - <<PY
import csv,glob
f=glob.glob('$OUT/trace/*/*kernel_trace.csv')[0]
d={}
for r in csv.DictReader(open(f)):
    k=r['Kernel_Name'].split('(')[0][-40:]
    d.setdefault(k,[]).append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in d.items():
    if len(v)>20: print(k,len(v),'total ms',round(sum(v)/1e3,2),'min',round(min(v)),'max',round(max(v)))
e=[v for k,v in d.items() if 'k_tile_expect' in k][0][-124:]
print('expect one evaluation', round(sum(e)/1e3,2),'ms', [round(x) for x in e[:12]], '...', [round(x) for x in e[-4:]])
PY
