#!/bin/bash
# which kernels the option-free tests (automatic selection only) and the whole GPU suite launch: rocprofv3 kernel-trace statistics
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
mkdir -p gpurun_out/r6reach
cd /tmp && export TMPDIR=/tmp
AUTO="tests/test_gpu_bench.py tests/test_gpu_distributed.py tests/test_gpu_flows.py tests/test_gpu_nccl.py tests/test_reference_quccsd.py tests/test_reference_stack.py tests/test_reference_traces.py tests/test_encodings.py"
REST="tests/test_gpu_abi.py tests/test_gpu_kernels.py tests/test_gpu_sector.py tests/test_gpu_tile.py tests/test_gpu_fullsize.py"
cd "$R"
SECONDS=0
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/reach_auto -- python3 -m pytest $AUTO -q -m gpu -p no:cacheprovider > gpurun_out/r6reach/auto.log 2>&1
echo "auto rc=$? seconds=$SECONDS"; grep -E "passed|failed" gpurun_out/r6reach/auto.log | tail -1
SECONDS=0
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/reach_rest -- python3 -m pytest $REST -q -m gpu -p no:cacheprovider > gpurun_out/r6reach/rest.log 2>&1
echo "rest rc=$? seconds=$SECONDS"; grep -E "passed|failed" gpurun_out/r6reach/rest.log | tail -1
python3 tools/kernel_reach.py gpurun_out/r6reach/kernel_reach.json option_free_tests=/tmp/reach_auto option_setting_tests=/tmp/reach_rest bench=profiles/r6a/kernel_stats.csv
du -sh /tmp/reach_auto /tmp/reach_rest; rm -rf /tmp/reach_auto /tmp/reach_rest
