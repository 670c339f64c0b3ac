#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r6h
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_flows.py tests/test_gpu_abi.py tests/test_reference_traces.py -x -q -m gpu > gpurun_out/r6h/t1.log 2>&1
grep -E "passed|failed|Error" gpurun_out/r6h/t1.log | tail -5
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_sector.py -x -q -m gpu -k "adapt or screen or exp" > gpurun_out/r6h/t2.log 2>&1
grep -E "passed|failed|Error" gpurun_out/r6h/t2.log | tail -5
timeout 600 python tools/exp_adapt_breakdown.py > gpurun_out/r6h/adapt_breakdown.log 2>&1
head -3 gpurun_out/r6h/adapt_breakdown.log; tail -3 gpurun_out/r6h/adapt_breakdown.log
