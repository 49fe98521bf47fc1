#!/bin/bash
# gpurun --timeout 1500 -- 'bash tools/gpu_runs/soak.sh 10'   -> gpurun_out/soak/soak.log
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/soak; mkdir -p $O
timeout $(( ${1:-5} * 60 + 300 )) python3 tests/soak.py ${1:-5} > $O/soak.log 2>&1; echo "rc=$?"; tail -n 3 $O/soak.log
