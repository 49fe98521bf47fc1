#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 2400 python -m pytest tests/test_gpu_dist.py -x -q -m gpu -k "dry_ranks or two_ranks" 2>&1 | tail -20 | tee gpurun_out/r06_dry_ranks_tests.txt
