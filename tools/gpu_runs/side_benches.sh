#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/side
mkdir -p $O
timeout 600 python3 tools/sweep.py > $O/sweep.jsonl 2> $O/sweep.err
timeout 300 python3 tools/bench_prove.py 18 234 5 1 1 > $O/prove.json 2> $O/prove.err
timeout 300 python3 tools/bench_quotient_ed25519.py 18 5 1 > $O/quotient.json 2> $O/quotient.err
timeout 300 python3 tools/bench_pcie.py > $O/pcie.json 2> $O/pcie.err
for db in 16 17 19 20; do timeout 300 python3 tools/bench_prove.py $db 234 3 0 1 >> $O/prove_scaling.jsonl 2>> $O/prove.err; done
cat $O/sweep.jsonl
tail -c 1500 $O/prove.json; tail -c 800 $O/quotient.json
