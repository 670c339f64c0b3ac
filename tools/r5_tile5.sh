#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python -m pytest tests/test_gpu_tile.py tests/test_gpu_fullsize.py tests/test_gpu_kernels.py tests/test_gpu_distributed.py tests/test_gpu_nccl.py tests/test_gpu_abi.py -q -m gpu -x -k "tile or 30_qubit or 32_qubit or expect or shard or bilinear or rccl or census" 2>&1 | tail -3
export OVQE_LIB=testing
for o in "" "tile_unsplit=0"; do echo "== $o"; OVQE_OPTIONS="$o" python tools/exp_shard_h.py 29 2>&1 | tail -1 | cut -c1-200; done
python tools/exp_shard_h.py 31 2>&1 | tail -1 | cut -c1-260
