#!/bin/bash
# M3: molecule-shaped UCCSD energy evaluation at 24 qubits on the streaming kernels under rocprofv3:
# kernel-trace stats + separate FETCH_SIZE / WRITE_SIZE passes.  The workload is built once and pickled in /tmp.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-prof_n24}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/tools/exp_mid.py 12 5 --cache /tmp/wl24.pkl > $OUT/plain.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/tools/exp_mid.py 12 5 --cache /tmp/wl24.pkl > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $R/tools/exp_mid.py 12 5 --cache /tmp/wl24.pkl > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $R/tools/exp_mid.py 12 5 --cache /tmp/wl24.pkl > $OUT/pmc_write.log 2>&1
tail -3 $OUT/plain.log
