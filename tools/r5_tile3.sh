#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export OVQE_LIB=testing
python -m pytest tests/test_gpu_tile.py tests/test_gpu_fullsize.py -q -m gpu -x -k "tile or 30_qubit or 32_qubit" 2>&1 | tail -2
for o in "" "expect_dense=0" "tile_flat=0" "tile_flat=0,expect_dense=0"; do echo "== $o"; OVQE_OPTIONS="$o" python tools/exp_shard_h.py ${1:-29} 2>&1 | tail -1 | cut -c1-200; done
