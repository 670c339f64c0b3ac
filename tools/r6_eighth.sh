#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r6h
export OVQE_LIB=testing
for v in "" "sector_bits=15" "sector_bits=15 sector_reg_threads=512" "sector_bits=16 sector_reg_threads=512" "sector_bits=16 sector_reg_threads=1024" "sector_bits=13" "sector_bits=13 sector_reg_threads=128"; do
  echo "== $v"
  timeout 300 python tools/exp_quccsd_reg.py reps=8 grad=3 $v 2>&1 | grep -v amdgpu.ids | cut -c1-420
done | tee gpurun_out/r6h/quccsd_tile_bits.log
