#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02t1; mkdir -p $O
timeout 2400 python3 -m pytest tests/test_gpu_prove.py tests/test_gpu_plonk.py -x -q -m gpu -k "2e10 or all_25 or known_answer" --durations=8 > $O/t1.log 2>&1
echo "rc=$?" >> $O/t1.log
tail -25 $O/t1.log
