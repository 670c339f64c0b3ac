#!/bin/bash
# round 6, first GPU call: the cross-shard tile kernels against the oracle, then the sharded leg (one shard at 31 qubits), the dry
# 8-rank leg at 34 qubits and the world-2 single-device run at 30-qubit shards
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r6
timeout 900 python -m pytest tests/test_gpu_distributed.py -x -q > gpurun_out/r6/test_gpu_distributed.log 2>&1
echo "test_gpu_distributed rc=$?" | tee -a gpurun_out/r6/summary.txt
tail -5 gpurun_out/r6/test_gpu_distributed.log
timeout 900 python bench.py --steps 3 --warmup 1 --no-roofline --no-cpu --no-extra > gpurun_out/r6/bench_n1_sharded.log 2>&1
echo "bench n1 rc=$?" | tee -a gpurun_out/r6/summary.txt
tail -1 gpurun_out/r6/bench_n1_sharded.log
cp gpurun_out/bench_extra.json gpurun_out/r6/bench_extra_n1.json
OVQE_BENCH_BACKEND=gloo OVQE_BENCH_SINGLE_DEVICE=1 timeout 1500 python bench.py --gpus 2 --steps 2 --warmup 1 --batch 4096 --no-roofline --no-cpu --no-extra --sharded-qubits 30 > gpurun_out/r6/bench_w2_single_device.log 2>&1
echo "bench w2 rc=$?" | tee -a gpurun_out/r6/summary.txt
tail -1 gpurun_out/r6/bench_w2_single_device.log
cp gpurun_out/bench_extra.json gpurun_out/r6/bench_extra_w2.json
