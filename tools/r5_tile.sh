#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python -m pytest tests/test_gpu_tile.py tests/test_gpu_fullsize.py -q -m gpu -x -k "tile or 30_qubit or 32_qubit" 2>&1 | tail -3
OVQE_LIB=testing python tools/exp_shard_h.py 29 2>&1 | tail -1 | cut -c1-300
bash tools/r5_tilexp.sh "" flat2 | tail -2 | cut -c1-500
