#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export OVQE_LIB=testing
python -m pytest tests/test_gpu_abi.py -q -m gpu -x 2>&1 | tail -2
for o in "" "sector_reg_adjoint_threads=512" "sector_reg_threads=512" "sector_reg_threads=128"; do
  echo "== $o"; python tools/exp_quccsd_reg.py reps=6 grad=4 $o 2>&1 | tail -2 | cut -c1-200
done
python tools/fuzz_sector.py 12 5 2>&1 | tail -1
