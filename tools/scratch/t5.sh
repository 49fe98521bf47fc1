#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02t5; mkdir -p $O
timeout 2400 python3 -m pytest tests/test_gpu_merkle.py tests/test_golden.py tests/test_gpu_field.py -x -q -m gpu > $O/t5.log 2>&1
echo "rc=$?" >> $O/t5.log
tail -6 $O/t5.log
python3 tools/bench_poseidon.py
timeout 600 python3 bench.py --no-cpu --steps 3 --warmup 1 > $O/bench.json 2>$O/bench.err; python3 -c "import json; d=json.loads(open('$O/bench.json').read()); print(d['extra'].get('commit_ms'), d['extra'].get('commit_ms_without_leaf_major_copy'), d['extra']['prove']['prove_ms'], d['extra']['prove']['stage_ms'])"
