#!/bin/bash
# the default bench (the driver's command) and the whole GPU suite on the final build
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r6f
SECONDS=0
timeout 900 python bench.py > gpurun_out/r6f/bench.log 2>&1
echo "bench rc=$? seconds=$SECONDS" | tee -a gpurun_out/r6f/summary.txt
tail -1 gpurun_out/r6f/bench.log > gpurun_out/r6f/bench.json
cp gpurun_out/bench_extra.json gpurun_out/r6f/bench_extra.json
tail -1 gpurun_out/r6f/bench.log | cut -c1-1200
SECONDS=0
timeout 2400 python -m pytest tests/ -x -q -m gpu --durations=25 > gpurun_out/r6f/gpu_suite.log 2>&1
echo "gpu suite rc=$? seconds=$SECONDS" | tee -a gpurun_out/r6f/summary.txt
grep -E "passed|failed" gpurun_out/r6f/gpu_suite.log | tail -2
grep -A27 "slowest" gpurun_out/r6f/gpu_suite.log | head -30
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
