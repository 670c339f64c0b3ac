#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
python -m pytest "$@" -q -m gpu -x 2>&1 | grep -v "^$" | tail -40
