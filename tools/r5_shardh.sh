#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export OVQE_LIB=testing
for o in "" "tile_flat=0" "tile_flat=2"; do
  echo "== $o"; OVQE_OPTIONS="$o" python tools/exp_shard_h.py ${1:-29} 2>&1 | tail -1 | cut -c1-300
done
