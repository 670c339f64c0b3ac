#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r6c
timeout 1500 python -m pytest tests/test_gpu_distributed.py -x -q > gpurun_out/r6c/test_gpu_distributed.log 2>&1
echo "test_gpu_distributed rc=$?" | tee -a gpurun_out/r6c/summary.txt
tail -4 gpurun_out/r6c/test_gpu_distributed.log | grep -v Gloo
timeout 1200 python -m pytest tests/test_gpu_fullsize.py -x -q -k "quccsd" > gpurun_out/r6c/test_gpu_fullsize_quccsd.log 2>&1
echo "test_gpu_fullsize quccsd rc=$?" | tee -a gpurun_out/r6c/summary.txt
tail -4 gpurun_out/r6c/test_gpu_fullsize_quccsd.log
OVQE_LIB=testing timeout 900 python tools/exp_real_shard.py 31 > gpurun_out/r6c/exp_real_shard.log 2>&1
echo "exp_real_shard rc=$?" | tee -a gpurun_out/r6c/summary.txt
grep variant gpurun_out/r6c/exp_real_shard.log
