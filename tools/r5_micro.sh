#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/lds_chain tools/micro/lds_chain.hip && /tmp/lds_chain
for o in "$@"; do
  echo "== $o"
  OVQE_LIB=testing python tools/exp_streams.py 12 5 ${o//,/ } 2>&1 | grep -v "amdgpu.ids\|without ops" | cut -c1-400
done
