#!/bin/bash
# where the time to the first N2 UCCSD energy from the sector tables goes: phases (sector_debug 4) and kernels (rocprofv3)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
OVQE_LIB=testing python tools/exp_setup_n2.py uccsd sector_debug=4 2>&1 | grep -v amdgpu.ids | cut -c1-300
OVQE_LIB=testing python tools/exp_setup_n2.py uccsd 2>&1 | grep "uccsd:" | cut -c1-300
bash tools/profile_any.sh r5_setup_prof tools/exp_setup_n2.py uccsd 2>&1 | cut -c1-200
sed -n 2,40p gpurun_out/r5_setup_prof/kernel_stats.csv | awk -F, '{printf "%s calls %s total_ns %s\n", substr($1,1,60), $2, $3}'
