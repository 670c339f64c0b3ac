#!/bin/bash
# the whole GPU suite with per-test durations (the driver's round-end step has 1200 s for it); then the library variants
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r6suite
SECONDS=0
timeout 2400 python -m pytest tests/ -x -q -m gpu --durations=40 > gpurun_out/r6suite/gpu_suite.log 2>&1
echo "gpu suite rc=$? seconds=$SECONDS" | tee gpurun_out/r6suite/summary.txt
grep -E "passed|failed" gpurun_out/r6suite/gpu_suite.log | tail -3
grep -A45 "slowest" gpurun_out/r6suite/gpu_suite.log | head -50
for v in openvqe_amd/lib/libovqe_sv.so openvqe_amd/lib/variants/libovqe_sv_compress.so openvqe_amd/lib/variants/libovqe_sv_stripall.so; do
  timeout 900 python tools/exp_lib_variants.py $v 2>&1 | tail -2 | tee -a gpurun_out/r6suite/lib_variants.log
done
