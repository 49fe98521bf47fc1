#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/ntt_small; mkdir -p $O
timeout 1500 python3 -m pytest tests/test_gpu_ntt.py tests/test_golden.py -x -q -m gpu > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log; tail -n 3 $O/tests.log
SIZES=16,17,18,19,20 python3 tools/gpu_runs/ntt_time_sizes.py
SIZES=17,18,19 python3 tools/gpu_runs/lde_time.py
