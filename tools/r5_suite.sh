#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/r5suite
( time python -m pytest tests/ -x -q -m gpu --durations=25 ) > gpurun_out/r5suite/pytest.log 2>&1
grep -v "COBYLA\|NFVALS\|X =\|^$\|^   " gpurun_out/r5suite/pytest.log | tail -45
