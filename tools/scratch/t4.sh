#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02t4; mkdir -p $O
timeout 2400 python3 -m pytest tests/test_gpu_merkle.py tests/test_golden.py tests/test_gpu_prove.py tests/test_gpu_fri.py -x -q -m gpu > $O/t4.log 2>&1
echo "rc=$?" >> $O/t4.log
tail -15 $O/t4.log
for pl in 1 0 1 0; do PLONKY2_COMMIT_PIPELINE=$pl timeout 600 python3 bench.py --no-cpu --steps 3 --warmup 1 > $O/bench_pl$pl.json 2>$O/bench_pl$pl.err; python3 -c "import json; d=json.loads(open('$O/bench_pl$pl.json').read()); print('pipeline $pl', d['extra'].get('commit_ms'), d['extra'].get('commit_ms_without_leaf_major_copy'), d['extra']['prove']['prove_ms'], d['extra']['prove']['stage_ms'])"; done
