#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python -m pytest tests/test_gpu_sector.py tests/test_gpu_fullsize.py tests/test_reference_quccsd.py -q -m gpu -x -k "coset or quccsd or fuzz" 2>&1 | tail -3
OVQE_LIB=testing python tools/exp_setup_n2.py quccsd sector_debug=4 2>&1 | grep -v "^/opt" | tail -14
python tools/exp_quccsd_reg.py reps=5 grad=2 2>&1 | tail -2 | cut -c1-200
