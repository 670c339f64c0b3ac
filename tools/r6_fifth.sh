#!/bin/bash
# round 6: two and four REAL ranks on one GPU (gloo, host-staged exchange, compute sections in turn), then the round profile of the
# bench command (kernel-trace stats + separate FETCH_SIZE / WRITE_SIZE passes)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r6e
for W in 2 4; do
  Q=$((31 - (W == 2 ? 1 : 2)))
  OVQE_BENCH_BACKEND=gloo OVQE_BENCH_SINGLE_DEVICE=1 timeout 1500 python bench.py --gpus $W --steps 2 --warmup 1 --batch 4096 --no-roofline --no-cpu --no-extra --sharded-qubits $Q > gpurun_out/r6e/bench_w${W}_single_device.log 2>&1
  echo "bench w$W rc=$?" | tee -a gpurun_out/r6e/summary.txt
  tail -1 gpurun_out/r6e/bench_w${W}_single_device.log | cut -c1-1500
  cp gpurun_out/bench_extra.json gpurun_out/r6e/bench_extra_w${W}.json
done
bash tools/profile_bench.sh r6e/prof > gpurun_out/r6e/profile.log 2>&1
echo "profile rc=$?" | tee -a gpurun_out/r6e/summary.txt
python tools/summarize_profile.py gpurun_out/r6e/prof gpurun_out/r6e/summary_prof 2>&1 | tail -2
tail -1 gpurun_out/r6e/prof/trace.log | cut -c1-600
# keep the summaries, drop the raw traces (gpurun copies back at most 64 MiB)
cp gpurun_out/r6e/prof/trace.log gpurun_out/r6e/summary_prof/bench_profiled_run.log 2>/dev/null
tail -1 gpurun_out/r6e/prof/trace.log > gpurun_out/r6e/summary_prof/bench_profiled_run.json 2>/dev/null
rm -rf gpurun_out/r6e/prof
du -sh gpurun_out
