#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export OVQE_LIB=testing
export OVQE_OPTIONS="$1"
bash tools/profile_pmc_any.sh r5_tilexp_$2 "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY" tools/exp_shard_h.py 28 2>&1 | grep -i "tile_expect"
grep -v "^/opt" gpurun_out/r5_tilexp_$2/run.log | tail -1 | cut -c1-200
