#!/bin/bash
# LDS counters of the third sweep form, lanes of the rows in list order against arranged for the banks; then the sector tests in full
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/r5s
python -m pytest tests/test_gpu_sector.py -x -q -m gpu > gpurun_out/r5s/pytest_sector.log 2>&1
grep -v "^  File\|^    " gpurun_out/r5s/pytest_sector.log | tail -30 | cut -c1-250
export OVQE_LIB=testing
for a in 0 1; do
  bash tools/profile_pmc_any.sh r5s_pmc$a "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU" tools/exp_streams.py 12 5 sector_stream_arrange=$a > /dev/null 2>&1
  echo "== arrange $a"; grep "k_sector_sweep3\|k_sector_sweep2" gpurun_out/r5s_pmc$a/pmc_summary.txt | cut -c1-600
done
