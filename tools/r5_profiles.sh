#!/bin/bash
# round-5 profiles of the final build: configs[3] (QUCCSD, evaluation + gradient) and the bench command; summaries only (raw traces removed)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
bash tools/profile_quccsd24.sh r5_quccsd24 grad=3 > gpurun_out/r5_quccsd24.log 2>&1
tail -25 gpurun_out/r5_quccsd24.log | cut -c1-400
bash tools/profile_bench.sh prof_r5b > gpurun_out/prof_r5b.log 2>&1
python3 tools/summarize_profile.py gpurun_out/prof_r5b gpurun_out/r5b
grep -o '{"metric.*' gpurun_out/prof_r5b/trace.log | tail -1 > gpurun_out/r5b/bench_profiled_run.json
rm -rf gpurun_out/prof_r5b/trace gpurun_out/prof_r5b/pmc_fetch gpurun_out/prof_r5b/pmc_write
head -12 gpurun_out/r5b/kernel_stats.csv | cut -c1-200
# third sweep form against the second on the same tables (24, 20, 16 qubits)
OVQE_LIB=testing python tools/exp_streams.py 12 5 10 4 8 3 2>&1 | grep -v "amdgpu.ids" > gpurun_out/r5_streams.log
grep -v "^m=" gpurun_out/r5_streams.log | cut -c1-200
