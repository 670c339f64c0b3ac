#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/r5m
OVQE_LIB=testing python tools/exp_latency_phases.py 2>&1 | grep -v "^/opt" | tail -12
( time python tools/fuzz_sector.py 40 77 ) > gpurun_out/r5m/fuzz.log 2>&1; tail -4 gpurun_out/r5m/fuzz.log
