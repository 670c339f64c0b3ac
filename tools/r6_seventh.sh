#!/bin/bash
# the ADAPT host profile, then the whole GPU suite with durations
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r6g
timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "real_state_option or planned_cross" 2>&1 | tail -5
timeout 600 python tools/prof_adapt_n2.py > gpurun_out/r6g/prof_adapt.log 2>&1
head -60 gpurun_out/r6g/prof_adapt.log
SECONDS=0
timeout 2400 python -m pytest tests/ -q -m gpu --durations=40 > gpurun_out/r6g/gpu_suite.log 2>&1
echo "gpu suite rc=$? seconds=$SECONDS" | tee -a gpurun_out/r6g/summary.txt
grep -E "passed|failed" gpurun_out/r6g/gpu_suite.log | tail -2
