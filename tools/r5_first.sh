#!/bin/bash
export OVQE_LIB=testing   # the measurement options these scripts pass exist in the -DOVQE_TESTING build only
# round 5, first GPU call: the bench tests (compact line, watchdog, energy check), the default bench run, barrier experiment
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/r5a
python -m pytest tests/test_gpu_bench.py tests/test_gpu_nccl.py -x -q -m gpu > gpurun_out/r5a/pytest_bench.log 2>&1
tail -5 gpurun_out/r5a/pytest_bench.log
( time python bench.py ) > gpurun_out/r5a/bench.log 2>&1
tail -c 4500 gpurun_out/r5a/bench.log
cp gpurun_out/bench_extra.json gpurun_out/r5a/bench_extra.json
for o in "" "sector_sweep_dbg=4" "sector_sweep_dbg=1" "sector_sweep_dbg=2" "sector_sweep_dbg=3"; do
  echo "== $o"; python tools/exp_quccsd_reg.py reps=8 $o 2>&1 | tail -1 | cut -c1-400
done
