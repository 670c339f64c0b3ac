#!/bin/bash
export OVQE_LIB=testing   # the measurement options these scripts pass exist in the -DOVQE_TESTING build only
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for o in "sector_reg_b5=1" "sector_reg_b5=0"; do
bash tools/profile_pmc_any.sh r5_b5_$o "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY" tools/exp_quccsd_reg.py reps=4 $o 2>&1 | grep "k_sector_sweep_reg" | cut -c1-500
done
