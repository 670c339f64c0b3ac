#!/bin/bash
# rocprofv3 kernel trace of the sector-path gradient at 24 qubits; usage: tools/profile_sector_grad.sh <tag> [exp args]
R=${GRAFT_REPO_ROOT:-/root/repo}
tag=${1:-secgrad}; shift
OUT=$R/gpurun_out/$tag
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/tools/exp_sector_grad.py 12 5 "$@" > $OUT/run.log 2>&1
find $OUT/trace -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
find $OUT/trace -name "*kernel_trace.csv" -exec rm {} \;
grep -E "k_sector|k_sec_dot|k_sec_reduce_w|k_adjoint" $OUT/kernel_stats.csv | sed 's/(.*)"/"/' | cut -c1-150
grep -E "^sector|^max" $OUT/run.log
