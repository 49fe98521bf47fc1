#!/bin/bash
# round 5: differential campaign of the gate-kernel generator (tests/fuzz_gate_jit.py), two seeds
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05gatefuzz; mkdir -p $O
S1=${1:-$(date +%s)}; S2=$((S1 + 1))
for s in $S1 $S2; do
  timeout 1500 python3 tests/fuzz_gate_jit.py 40 $s > $O/seed_$s.log 2>&1; echo "seed $s rc=$?" | tee -a $O/seed_$s.log
  tail -n 2 $O/seed_$s.log; grep -h "FAIL\|MISMATCH\|Error\|error" $O/seed_$s.log | head -5
done
