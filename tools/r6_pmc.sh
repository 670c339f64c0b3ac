#!/bin/bash
# SQ counters of the tiled <H> kernels on the 31-qubit shard leg (one rank): what bounds k_tile_expect after round 6's changes
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r6h
bash tools/profile_pmc_any.sh r6h/sq "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_INST_LDS" tools/bench_sharded.py --qubits 31 > gpurun_out/r6h/sq.log 2>&1
grep -E "k_tile_expect|k_tile_diag|k_tile_sweep" gpurun_out/r6h/sq/pmc_summary.txt | cut -c1-700
bash tools/profile_pmc_any.sh r6h/sq2 "SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_WAVES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" tools/bench_sharded.py --qubits 31 > gpurun_out/r6h/sq2.log 2>&1
grep -E "k_tile_expect|k_tile_diag" gpurun_out/r6h/sq2/pmc_summary.txt | cut -c1-700
