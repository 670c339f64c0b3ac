#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/r5s
python -m pytest tests/test_gpu_abi.py -x -q -m gpu > gpurun_out/r5s/pytest_abi.log 2>&1
tail -3 gpurun_out/r5s/pytest_abi.log
OVQE_LIB=testing python tools/exp_setup_n2.py sector_debug=4 > gpurun_out/r5s/setup.log 2>&1
grep -v "^/opt" gpurun_out/r5s/setup.log | tail -80
