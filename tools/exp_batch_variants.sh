#!/bin/bash
# ovqe_energy_batch on the sector tables, 24 qubits, B = 64: states per tile x threads of the batched <H> kernel
R=${GRAFT_REPO_ROOT:-/root/repo}
for nb in 2 3; do for nt in 512 1024; do
  echo "== sector_batch_nb $nb sector_batch_threads $nt"
  python3 $R/tools/exp_sector_batch.py 12 5 --B=64 --opt=sector_batch_nb=$nb --opt=sector_batch_threads=$nt 2>&1 | grep -E "serial|B="
done; done
