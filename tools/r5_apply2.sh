#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python -m pytest tests/test_gpu_sector.py tests/test_gpu_fullsize.py -q -m gpu -x -k "gradient or ground_state or screen or fuzz or adapt or vqe or quccsd" 2>&1 | tail -3
export OVQE_LIB=testing
for o in "" "sector_apply_seq=0"; do echo "== $o"; python tools/exp_quccsd_reg.py reps=4 grad=4 $o 2>&1 | tail -1 | cut -c1-140; done
python tools/exp_sector_grad.py 12 5 --sector-only 2>&1 | tail -2 | cut -c1-200
