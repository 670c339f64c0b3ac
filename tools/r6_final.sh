#!/bin/bash
# final build of round 6: the round profile of the bench command (kernel statistics + FETCH / WRITE passes), the default bench (the driver's
# command), the whole GPU suite and the smoke call
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r6f
bash tools/profile_bench.sh r6f/prof > gpurun_out/r6f/profile.log 2>&1
echo "profile rc=$?" | tee -a gpurun_out/r6f/summary.txt
python tools/summarize_profile.py gpurun_out/r6f/prof gpurun_out/r6f/summary_prof 2>&1 | tail -2
tail -1 gpurun_out/r6f/prof/trace.log > gpurun_out/r6f/summary_prof/bench_profiled_run.json 2>/dev/null
rm -rf gpurun_out/r6f/prof
SECONDS=0
timeout 900 python bench.py > gpurun_out/r6f/bench.log 2>&1
echo "bench rc=$? seconds=$SECONDS" | tee -a gpurun_out/r6f/summary.txt
tail -1 gpurun_out/r6f/bench.log > gpurun_out/r6f/bench.json
cp gpurun_out/bench_extra.json gpurun_out/r6f/bench_extra.json
tail -1 gpurun_out/r6f/bench.log | cut -c1-1200
SECONDS=0
timeout 2400 python -m pytest tests/ -x -q -m gpu --durations=25 > gpurun_out/r6f/gpu_suite.log 2>&1
echo "gpu suite rc=$? seconds=$SECONDS" | tee -a gpurun_out/r6f/summary.txt
grep -E "passed|failed" gpurun_out/r6f/gpu_suite.log | tail -2
grep -A27 "slowest" gpurun_out/r6f/gpu_suite.log | head -30 > gpurun_out/r6f/gpu_suite_durations.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
du -sh gpurun_out
