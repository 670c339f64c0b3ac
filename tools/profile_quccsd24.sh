#!/bin/bash
# configs[3] as written (N2 / cc-pVDZ (10e,12o) QUCCSD gate list, 24 qubits): rocprofv3 kernel trace + stats, then separate PMC passes
# (FETCH_SIZE, WRITE_SIZE) of tools/exp_quccsd_reg.py; summaries under gpurun_out/<tag>/ (copy into profiles/)
# usage: tools/profile_quccsd24.sh <tag> [option=value ...]
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=${1:-r4_quccsd24}; shift
OUT=$R/gpurun_out/$tag
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/tools/exp_quccsd_reg.py reps=12 "$@" > $OUT/run.log 2>&1
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
rm -rf $OUT/trace
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -- python3 $R/tools/exp_quccsd_reg.py reps=4 "$@" > $OUT/run_$c.log 2>&1
done
# the SQ block's eight slots in one pass: LDS / VALU instruction counts and busy cycles, waits, bank conflicts
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace --output-format csv -d $OUT/pmc_SQ -- python3 $R/tools/exp_quccsd_reg.py reps=4 "$@" > $OUT/run_SQ.log 2>&1
python3 - $OUT > $OUT/pmc_summary.txt <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:60]
        a = acc[k][r["Counter_Name"]]
        a[0] += 1; a[1] += float(r["Counter_Value"])
print("kernel, counter: (dispatches, mean raw value per dispatch [KiB for FETCH_SIZE / WRITE_SIZE; SQ_* cycle counters in quad-cycles summed over the waves])")
for k, d in sorted(acc.items(), key=lambda kv: -sum(v[1] for v in kv[1].values()))[:14]:
    print(k, {c: (n, round(v / n, 1)) for c, (n, v) in d.items()})
PY
rm -rf $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_SQ
python3 - $OUT/kernel_stats.csv <<'PY'
import csv, sys
for i, r in enumerate(csv.DictReader(open(sys.argv[1]))):
    if i < 12: print(f"{r['Name'][:70]:70s} calls {r['Calls']:>6s} total_ms {float(r['TotalDurationNs'])/1e6:9.3f} avg_us {float(r['AverageNs'])/1e3:10.2f} {r['Percentage']}%")
PY
cat $OUT/pmc_summary.txt
grep -v "^/opt" $OUT/run.log | tail -3
