#!/bin/bash
export OVQE_LIB=testing   # the measurement options these scripts pass exist in the -DOVQE_TESTING build only
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/r5b
python -m pytest tests/test_gpu_bench.py -x -q -m gpu -k "stalled or record" > gpurun_out/r5b/pytest_bench.log 2>&1
tail -3 gpurun_out/r5b/pytest_bench.log
python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "quccsd" > gpurun_out/r5b/pytest_quccsd.log 2>&1
tail -3 gpurun_out/r5b/pytest_quccsd.log
python tools/fuzz_sector.py 60 5 > gpurun_out/r5b/fuzz.log 2>&1
tail -3 gpurun_out/r5b/fuzz.log
for o in "" "sector_reg_runs=0" "sector_sweep_dbg=5" "sector_sweep_dbg=5 sector_reg_runs=0" "sector_sweep_dbg=4"; do
  echo "== $o"; python tools/exp_quccsd_reg.py reps=8 $o 2>&1 | tail -1 | cut -c1-330
done
for o in "" "sector_reg_runs=0"; do echo "== grad $o"; python tools/exp_quccsd_reg.py reps=4 grad=5 $o 2>&1 | tail -1 | cut -c1-300; done
