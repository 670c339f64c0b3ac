#!/bin/bash
export OVQE_LIB=testing   # the measurement options these scripts pass exist in the -DOVQE_TESTING build only
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python -m pytest "$@" -q -m gpu -x 2>&1 | grep -v "^$" | tail -40
