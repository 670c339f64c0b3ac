#!/bin/bash
# which HIP API call / kernel of a run takes more than a few milliseconds: rocprofv3 --hip-trace --kernel-trace of "$@" (the
# program directly after --), then the records above 5 ms;  usage: tools/trace_slow_calls.sh <name> python3 tools/... args
name=$1; shift
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
out=$R/gpurun_out/$name
mkdir -p $out
rocprofv3 --hip-trace --kernel-trace --output-format csv -d /tmp/trace_$name -o t -- "$@" > $out/run.log 2>&1
python3 - /tmp/trace_$name >> $out/slow.txt <<'PY'
import csv, sys, glob
d = sys.argv[1]
api = list(csv.DictReader(open(glob.glob(d + '/**/*hip_api_trace.csv', recursive=True)[0])))
ker = list(csv.DictReader(open(glob.glob(d + '/**/*kernel_trace.csv', recursive=True)[0])))
kname = {r['Correlation_Id']: r for r in ker}
t0 = min(int(r['Start_Timestamp']) for r in api)
def show(r, mark):
    dur = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
    k = kname.get(r['Correlation_Id'])
    extra = ''
    if k:
        extra = f"  -> {k['Kernel_Name'][:70]} (kernel {(int(k['End_Timestamp']) - int(k['Start_Timestamp'])) / 1e3:.1f} us, starts {(int(k['Start_Timestamp']) - int(r['Start_Timestamp'])) / 1e6:.2f} ms after the call, scratch {k['Scratch_Size']})"
    print(f"{mark} {(int(r['Start_Timestamp']) - t0) / 1e6:10.2f} ms  {dur:8.2f} ms  {r['Function']}{extra}")
for i, r in enumerate(api):
    if int(r['End_Timestamp']) - int(r['Start_Timestamp']) > 20e6 and (int(r['Start_Timestamp']) - t0) > 1.5e9:
        for j in range(max(0, i - 6), min(len(api), i + 3)):
            show(api[j], '>>' if j == i else '  ')
        print()
PY
