#!/bin/bash
# rocprofv3 of the sector path at 24 qubits (tools/exp_sector.py 12 5): kernel-trace stats, then separate FETCH_SIZE /
# WRITE_SIZE passes.  usage: tools/profile_sector.sh <tag> [--pmc] [extra exp_sector.py args]
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=${1:-sector}; shift
pmc=0; if [ "$1" = "--pmc" ]; then pmc=1; shift; fi
OUT=$R/gpurun_out/$tag
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/tools/exp_sector.py 12 5 "$@" > $OUT/run.log 2>&1
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
find $OUT/trace -name "*kernel_trace.csv" -exec rm {} \;
if [ $pmc = 1 ]; then
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -- python3 $R/tools/exp_sector.py 12 5 --sector-only "$@" > $OUT/pmc_$c.log 2>&1
    python3 - $OUT/pmc_$c $c >> $OUT/pmc_summary.txt <<'PY'
import csv, glob, sys, collections
d, c = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == c:
            k = r["Kernel_Name"].split("(")[0][:60]
            acc[k][0] += 1; acc[k][1] += float(r["Counter_Value"])
for k, (n, v) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:12]:
    print(f"{c},{k},{n},{v/n:.1f}")
PY
    rm -rf $OUT/pmc_$c
  done
  cat $OUT/pmc_summary.txt
fi
head -16 $OUT/kernel_stats.csv | cut -c1-70,180-320
grep -E "^sector|^E " $OUT/run.log | cut -c1-200
