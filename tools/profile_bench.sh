#!/bin/bash
# Round profile: kernel-trace stats + separate PMC passes (FETCH_SIZE / WRITE_SIZE) of the bench command.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-prof}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --no-cpu"
if [ -z "$PMC_ONLY" ]; then
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py $ARGS > $OUT/trace.log 2>&1
fi
# counter passes: same command without the side workloads (--no-extra; tens of thousands of small dispatches that
# counter collection serialises) — the timed step and the 30-qubit roofline sweeps are identical
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $R/bench.py $ARGS --no-extra > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $R/bench.py $ARGS --no-extra > $OUT/pmc_write.log 2>&1
find $OUT -name "*.csv" | head -30
