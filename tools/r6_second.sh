#!/bin/bash
# round 6, second GPU call: partitioned API + real-amplitude shards against the oracle, then the bench's sharded block (real-state leg)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r6b
timeout 1500 python -m pytest tests/test_gpu_distributed.py -x -q > gpurun_out/r6b/test_gpu_distributed.log 2>&1
echo "test_gpu_distributed rc=$?" | tee -a gpurun_out/r6b/summary.txt
tail -15 gpurun_out/r6b/test_gpu_distributed.log | grep -v Gloo
timeout 900 python -m pytest tests/test_gpu_bench.py tests/test_gpu_nccl.py -x -q > gpurun_out/r6b/test_gpu_bench.log 2>&1
echo "test_gpu_bench rc=$?" | tee -a gpurun_out/r6b/summary.txt
tail -5 gpurun_out/r6b/test_gpu_bench.log | grep -v Gloo
timeout 900 python bench.py --steps 3 --warmup 1 --no-roofline --no-extra > gpurun_out/r6b/bench_n1.log 2>&1
echo "bench n1 rc=$?" | tee -a gpurun_out/r6b/summary.txt
tail -1 gpurun_out/r6b/bench_n1.log
cp gpurun_out/bench_extra.json gpurun_out/r6b/bench_extra_n1.json
