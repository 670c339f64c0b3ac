#!/bin/bash
# third sweep form (per-wave streams): timings against the second form on the same tables; arguments: option sets to run ("a=1,b=2")
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/r5s
for o in "$@"; do
  echo "== $o"
  OVQE_LIB=testing python tools/exp_streams.py 12 5 ${o//,/ } 2>&1 | grep -v "amdgpu.ids" | cut -c1-300
done 2>&1 | tee gpurun_out/r5s/streams_waves.log
if [ -z "$QUICK" ]; then
OVQE_LIB=testing python tools/exp_streams.py 10 4 8 3 11 5 2>&1 | grep -v "amdgpu.ids\|without ops" | tee gpurun_out/r5s/streams.log | cut -c1-300
python -m pytest tests/test_gpu_sector.py -x -q -m gpu > gpurun_out/r5s/pytest_sector.log 2>&1
grep -v "^  File\|^    " gpurun_out/r5s/pytest_sector.log | tail -8 | cut -c1-250
fi
