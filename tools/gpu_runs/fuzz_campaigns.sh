#!/bin/bash
# The differential campaigns of tests/ (device against the oracle on randomly shaped cases) with fresh seeds.
#   gpurun -- 'bash tools/gpu_runs/fuzz_campaigns.sh [seed]'   -> gpurun_out/fuzz/*.log
cd "$GRAFT_REPO_ROOT" || exit 1
SEED=${1:-$(date +%s)}
O=gpurun_out/fuzz; mkdir -p $O
timeout 900 python3 tests/fuzz_commit.py 80 $SEED > $O/commit_small.log 2>&1; echo "commit small rc=$?" | tee -a $O/commit_small.log
timeout 1500 python3 tests/fuzz_commit.py 45 $SEED large > $O/commit_large.log 2>&1; echo "commit large rc=$?" | tee -a $O/commit_large.log
timeout 1500 python3 tests/fuzz_prove.py 60 $SEED > $O/prove.log 2>&1; echo "prove rc=$?" | tee -a $O/prove.log
for f in commit_small commit_large prove; do tail -n 2 $O/$f.log; done
grep -h FAIL $O/*.log | head
