#!/bin/bash
# last check of the round: the driver's three commands on the final tree (bench line, GPU suite with -x, smoke)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r6z
SECONDS=0
timeout 900 python bench.py > gpurun_out/r6z/bench.log 2>&1
echo "bench rc=$? seconds=$SECONDS" | tee -a gpurun_out/r6z/summary.txt
tail -1 gpurun_out/r6z/bench.log > gpurun_out/r6z/bench.json
cp gpurun_out/bench_extra.json gpurun_out/r6z/bench_extra.json
tail -1 gpurun_out/r6z/bench.log | cut -c1-400
SECONDS=0
timeout 2400 python -m pytest tests/ -x -q -m gpu --durations=15 > gpurun_out/r6z/gpu_suite.log 2>&1
echo "gpu suite rc=$? seconds=$SECONDS" | tee -a gpurun_out/r6z/summary.txt
grep -E "passed|failed" gpurun_out/r6z/gpu_suite.log | tail -2
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
