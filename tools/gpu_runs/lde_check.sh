#!/bin/bash
# coset LDE / commit parity, then the LDE of 135 columns alone with per-kernel times, and the 64-polynomial timings
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/lde_check; mkdir -p $O
timeout 1500 python3 -m pytest tests/test_gpu_ntt.py tests/test_golden.py tests/test_gpu_merkle.py -x -q -m gpu > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log; tail -n 4 $O/tests.log
bash tools/gpu_runs/lde_135.sh
SIZES=16,18,20 python3 tools/gpu_runs/lde_time.py
