#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/r5final
( time python -m pytest tests/ -x -q -m gpu --durations=12 ) > gpurun_out/r5final/pytest.log 2>&1
grep -v "COBYLA\|NFVALS\|X =\|^$\|^   " gpurun_out/r5final/pytest.log | tail -22
( time python bench.py ) > gpurun_out/r5final/bench.log 2>&1
tail -c 3200 gpurun_out/r5final/bench.log
cp gpurun_out/bench_extra.json gpurun_out/r5final/bench_extra.json
