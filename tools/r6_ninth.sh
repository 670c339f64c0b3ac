#!/bin/bash
# final build: two and four REAL ranks on one GPU (gloo, host-staged exchange, compute sections in turn) at 30- / 29-qubit shards
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r6i
for W in 2 4; do
  Q=$((31 - (W == 2 ? 1 : 2)))
  OVQE_BENCH_BACKEND=gloo OVQE_BENCH_SINGLE_DEVICE=1 timeout 1500 python bench.py --gpus $W --steps 2 --warmup 1 --batch 4096 --no-roofline --no-cpu --no-extra --sharded-qubits $Q > gpurun_out/r6i/bench_w${W}_single_device.log 2>&1
  echo "bench w$W rc=$?" | tee -a gpurun_out/r6i/summary.txt
  tail -1 gpurun_out/r6i/bench_w${W}_single_device.log | cut -c1-1800
  cp gpurun_out/bench_extra.json gpurun_out/r6i/bench_extra_w${W}.json
done
