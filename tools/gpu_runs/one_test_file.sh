#!/bin/bash
# gpurun -- 'bash tools/gpu_runs/one_test_file.sh tests/test_gpu_merkle.py [-k expr]'
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/one_test; mkdir -p $O
timeout 1800 python3 -m pytest "$@" -x -q -m gpu > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log
tail -n 25 $O/tests.log
