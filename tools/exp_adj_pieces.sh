#!/bin/bash
# k_sector_adjoint piece by piece (option sector_adj_dbg) under rocprofv3: 24 qubits
R=${GRAFT_REPO_ROOT:-/root/repo}
for d in 0 1 2 3; do
  echo "== sector_adj_dbg $d"
  bash $R/tools/profile_sector_grad.sh adj_dbg$d --opt=sector_adj_dbg=$d 2>&1 | grep -E "k_sector_adjoint|k_sector_apply"
done
