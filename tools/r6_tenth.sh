#!/bin/bash
# the 31-qubit energy on record (bench.SHARDED_KNOWN) from the C oracle on the host: 32-GiB state, ~10 min of CPU
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r6j
grep -E "MemTotal|MemAvailable" /proc/meminfo; nproc
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
timeout 3000 python tests/oracle_sharded_energy.py 31 64 1000 2>&1 | tee gpurun_out/r6j/oracle_31_qubits.log
