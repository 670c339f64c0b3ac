#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r6h
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python -m pytest tests/test_gpu_sector.py tests/test_gpu_abi.py -x -q -m gpu -k "adjoint or gradient or abi or forms or one_wave" > gpurun_out/r6h/t1.log 2>&1
grep -E "passed|failed|Error" gpurun_out/r6h/t1.log | tail -3
