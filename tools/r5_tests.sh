#!/bin/bash
# run a selection of GPU tests: arguments = pytest arguments
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/r5t
python -m pytest "$@" -q -m gpu --durations=15 > gpurun_out/r5t/pytest.log 2>&1
tail -40 gpurun_out/r5t/pytest.log
