#!/bin/bash
# round 5: the ed25519 quotient on the final code (timing beside the interpreter and the reference symbol; without gate constraints)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05qlast; mkdir -p $O
timeout 600 python3 tools/bench_quotient_ed25519.py 18 7 1 > $O/quotient.json 2> $O/quotient.err; tail -c 1000 $O/quotient.json
