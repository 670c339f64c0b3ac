#!/bin/bash
# rocprofv3 PMC pass (counters only, with the kernel trace) of an arbitrary tool script, summed per kernel:
# tools/profile_pmc_any.sh <tag> "<COUNTER1 COUNTER2 ...>" <script.py> [args]
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=$1; shift
ctrs=$1; shift
OUT=$R/gpurun_out/$tag
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $OUT/pmc -- python3 $R/"$@" > $OUT/run.log 2>&1
python3 - $OUT/pmc > $OUT/pmc_summary.txt <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:50]
        a = acc[k][r["Counter_Name"]]
        a[0] += 1; a[1] += float(r["Counter_Value"])
for k, d in sorted(acc.items(), key=lambda kv: -sum(v[1] for v in kv[1].values()))[:40]:
    print(k, {c: (n, round(v / n, 1)) for c, (n, v) in d.items()})
PY
rm -rf $OUT/pmc
cat $OUT/pmc_summary.txt
grep -v "^/opt" $OUT/run.log | grep -E "renumber|kernel" | head
