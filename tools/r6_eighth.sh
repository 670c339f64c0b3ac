#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r6h
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_flows.py tests/test_gpu_abi.py tests/test_reference_traces.py -x -q -m gpu > gpurun_out/r6h/t1.log 2>&1
grep -E "passed|failed|Error" gpurun_out/r6h/t1.log | tail -5
