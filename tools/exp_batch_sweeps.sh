#!/bin/bash
# circuit sweeps of a batch on the sector tables, 24 qubits, B = 64: workgroup size x scatter indices in LDS or not
R=${GRAFT_REPO_ROOT:-/root/repo}
for nt in 1024 512; do for dl in 1 0; do
  echo "== sector_batch_sweep_threads $nt sector_batch_dst_lds $dl"
  python3 $R/tools/exp_sector_batch.py 12 5 --B=64 --opt=sector_batch_sweep_threads=$nt --opt=sector_batch_dst_lds=$dl 2>&1 | grep -E "B="
done; done
