#!/bin/bash
# k_sector_apply with 512 / 1024 threads per workgroup under rocprofv3: 24 qubits
R=${GRAFT_REPO_ROOT:-/root/repo}
for d in 512 1024; do
  echo "== sector_apply_threads $d"
  bash $R/tools/profile_sector_grad.sh apply_nt$d --sector-only --opt=sector_apply_threads=$d 2>&1 | grep -E "k_sector_apply|^sector"
done
