#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02t7; mkdir -p $O
timeout 1500 python3 -m pytest tests/test_gpu_ntt.py -x -q -m gpu > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log
tail -8 $O/tests.log
timeout 600 python3 tools/sweep.py > $O/sweep.jsonl 2> $O/sweep.err
head -8 $O/sweep.jsonl
