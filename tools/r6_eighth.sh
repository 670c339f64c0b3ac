#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r6h
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in 1 0; do
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf$v -- python3 $R/tools/exp_quccsd_reg.py reps=4 grad=6 sector_apply_full=$v > /tmp/pf$v.log 2>&1
f=$(find /tmp/pf$v -name "*kernel_stats.csv" | head -1)
echo "== sector_apply_full=$v"; grep -E "k_sector_apply|k_sector_adjoint_reg|k_sector_sweep_reg|k_sector_expect" $f | sed 's/(.*)"/"/' | cut -c1-200
done | tee $R/gpurun_out/r6h/apply_full_kernels.log
rm -rf /tmp/pf0 /tmp/pf1
