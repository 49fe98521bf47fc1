#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/commit_quick; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_merkle.py -x -q -m gpu -k "commit or merkle_tree" > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log; tail -n 3 $O/tests.log
for rep in 1 2 3; do TAG=product python3 tools/gpu_runs/commit_time.py; done
