#!/bin/bash
# per-sweep work of the Hamiltonian tile cover beside the per-sweep kernel durations (diagnostics)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-prof_cover}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export OVQE_DEBUG_COVER=1
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $R/tools/exp_mid.py 12 5 --cache /tmp/wl24.pkl --once > $OUT/trace.log 2>&1
grep "cover sweep" $OUT/trace.log > $OUT/cover.txt
python3 - <<PY
import csv,glob,re
f=glob.glob("$OUT/trace/*/*kernel_trace.csv")[0]
d=[int(r["End_Timestamp"])-int(r["Start_Timestamp"]) for r in csv.DictReader(open(f)) if "k_tile_expect" in r["Kernel_Name"]]
cov=[list(map(int,re.findall(r"\d+",l))) for l in open("$OUT/cover.txt")]
n=len(cov)
d=d[-n:]
print("sweeps",n,"total ms",sum(d)/1e6)
for c,t in list(zip(cov,d))[:12]+list(zip(cov,d))[-5:]:
    print(c, "us", t/1e3, " pair-terms/us/tile", c[-1]/(t/1e3))
PY
