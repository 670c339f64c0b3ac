#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r6g
timeout 1500 python -m pytest tests/test_gpu_tile.py tests/test_gpu_kernels.py tests/test_gpu_distributed.py -x -q > gpurun_out/r6g/tests.log 2>&1
echo "tile+kernels+distributed rc=$?" | tee -a gpurun_out/r6g/summary.txt
grep -E "passed|failed|Error" gpurun_out/r6g/tests.log | tail -3
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -k "30_qubit or sparse_oracle or tiled" > gpurun_out/r6g/tests30.log 2>&1
echo "30-qubit tests rc=$?" | tee -a gpurun_out/r6g/summary.txt
timeout 900 python bench.py --steps 3 --warmup 1 --no-roofline --no-cpu --no-extra > gpurun_out/r6g/bench_n1.log 2>&1
echo "bench rc=$?" | tee -a gpurun_out/r6g/summary.txt
tail -1 gpurun_out/r6g/bench_n1.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps(d['sharded']))"
cp gpurun_out/bench_extra.json gpurun_out/r6g/bench_extra_n1.json
OVQE_LIB=testing timeout 600 python tools/exp_real_shard.py 31 > gpurun_out/r6g/exp_real_shard.log 2>&1
grep variant gpurun_out/r6g/exp_real_shard.log | cut -c1-200
