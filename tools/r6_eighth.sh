#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r6h
timeout 600 python tools/exp_adapt_breakdown.py > gpurun_out/r6h/adapt_breakdown.log 2>&1
head -3 gpurun_out/r6h/adapt_breakdown.log | cut -c1-200; grep -E "^prog (41|43|45|47|59)" gpurun_out/r6h/adapt_breakdown.log;  tail -3 gpurun_out/r6h/adapt_breakdown.log | cut -c1-200
timeout 1200 python -m pytest tests/test_gpu_sector.py tests/test_gpu_flows.py tests/test_reference_traces.py tests/test_gpu_fullsize.py -x -q -m gpu > gpurun_out/r6h/t1.log 2>&1
grep -E "passed|failed|Error" gpurun_out/r6h/t1.log | tail -5
