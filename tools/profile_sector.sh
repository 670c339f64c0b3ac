#!/bin/bash
# rocprofv3 kernel trace of the sector path at 24 qubits; usage: tools/profile_sector.sh <tag> [extra exp_sector.py args]
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=${1:-sector}; shift
OUT=$R/gpurun_out/$tag
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/tools/exp_sector.py 12 5 "$@" > $OUT/run.log 2>&1
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
find $OUT/trace -name "*kernel_trace.csv" -exec rm {} \;
head -24 $OUT/kernel_stats.csv
grep -E "^sector|^E " $OUT/run.log
