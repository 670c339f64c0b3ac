#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export OVQE_LIB=testing
python -m pytest tests/test_gpu_tile.py tests/test_gpu_fullsize.py -q -m gpu -x -k "tile or 30_qubit or 32_qubit" 2>&1 | tail -2
OVQE_OPTIONS="tile_flat=0" python -m pytest tests/test_gpu_tile.py tests/test_gpu_fullsize.py -q -m gpu -x -k "tile or 30_qubit or 32_qubit" 2>&1 | tail -2
for o in "" "tile_flat=0"; do echo "== $o"; OVQE_OPTIONS="$o" python tools/exp_shard_h.py 29 2>&1 | tail -1 | cut -c1-200; done
bash tools/r5_tilexp.sh "tile_flat=0" entries2 | tail -2 | cut -c1-500
