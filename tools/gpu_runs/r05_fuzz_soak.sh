#!/bin/bash
# round 5: the differential campaigns with a fresh seed (the new one against the C prover included) and an eight-minute soak
cd "$GRAFT_REPO_ROOT" || exit 1
SEED=${1:-$(date +%s)}
O=gpurun_out/r05fuzz; mkdir -p $O
echo "seed $SEED" > $O/seed.txt
timeout 900 python3 tests/fuzz_commit.py 80 $SEED > $O/commit_small.log 2>&1; echo "commit small rc=$?" | tee -a $O/commit_small.log
timeout 1500 python3 tests/fuzz_commit.py 45 $SEED large > $O/commit_large.log 2>&1; echo "commit large rc=$?" | tee -a $O/commit_large.log
timeout 1500 python3 tests/fuzz_prove.py 60 $SEED > $O/prove.log 2>&1; echo "prove rc=$?" | tee -a $O/prove.log
timeout 1800 python3 tests/fuzz_prove_c.py 120 $SEED > $O/prove_c.log 2>&1; echo "prove_c rc=$?" | tee -a $O/prove_c.log
timeout 900 python3 tests/soak.py 8 > $O/soak.log 2>&1; echo "soak rc=$?" | tee -a $O/soak.log
for f in commit_small commit_large prove prove_c soak; do tail -n 2 $O/$f.log; done
grep -h FAIL $O/*.log | head
