#!/bin/bash
export OVQE_LIB=testing   # the measurement options these scripts pass exist in the -DOVQE_TESTING build only
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python -m pytest tests/test_gpu_tile.py tests/test_gpu_fullsize.py tests/test_gpu_kernels.py tests/test_gpu_distributed.py tests/test_gpu_nccl.py -q -m gpu -x -k "tile or 30_qubit or 32_qubit or expect or shard or bilinear or rccl" 2>&1 | tail -3
python tools/exp_shard_h.py 31 2>&1 | tail -1 | cut -c1-260
python tools/exp_shard_h.py 28 2>&1 | tail -1 | cut -c1-260
