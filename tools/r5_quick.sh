#!/bin/bash
export OVQE_LIB=testing   # the measurement options these scripts pass exist in the -DOVQE_TESTING build only
# quick check of the regular sweeps: QUCCSD oracle test, fuzz, timings (evaluation + gradient) with option variants given as arguments
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/r5q
python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "quccsd" > gpurun_out/r5q/pytest_quccsd.log 2>&1
tail -2 gpurun_out/r5q/pytest_quccsd.log
python tools/fuzz_sector.py 40 11 > gpurun_out/r5q/fuzz.log 2>&1
tail -1 gpurun_out/r5q/fuzz.log
for o in "" "$@"; do
  echo "== $o"; python tools/exp_quccsd_reg.py reps=8 grad=4 $o 2>&1 | tail -2 | cut -c1-330
done
