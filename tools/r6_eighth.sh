#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r6h
SECONDS=0
timeout 2400 python -m pytest tests/ -q -m gpu --durations=30 > gpurun_out/r6h/gpu_suite.log 2>&1
echo "gpu suite rc=$? seconds=$SECONDS"
grep -E "passed|failed" gpurun_out/r6h/gpu_suite.log | tail -2
grep -E "^FAILED|^ERROR" gpurun_out/r6h/gpu_suite.log | head
grep -A12 "slowest" gpurun_out/r6h/gpu_suite.log | head -14
