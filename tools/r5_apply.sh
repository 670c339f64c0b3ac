#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export OVQE_LIB=testing
for o in "" "sector_h_dbg=8"; do echo "== $o"; python tools/exp_quccsd_reg.py reps=4 grad=4 $o 2>&1 | tail -1 | cut -c1-120; done
