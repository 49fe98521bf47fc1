#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/suite
mkdir -p $O
timeout 1500 python3 -m pytest tests -x -q -m gpu > $O/gpu_suite.log 2>&1
echo "tests rc=$?" >> $O/gpu_suite.log
timeout 900 python3 bench.py > $O/bench.json 2> $O/bench.err
timeout 300 python3 bench.py --gpus 2 --steps 5 --warmup 1 --no-commit --no-cpu > $O/bench_2ranks.json 2> $O/bench_2ranks.err
tail -4 $O/gpu_suite.log
python3 -c "import json; d=json.loads(open('$O/bench.json').read()); print(d['value'], d['roofline']['ms'], d['roofline']['frac'], d['extra'].get('commit_ms'), d['extra']['prove']['prove_ms'], d['cpu_baseline'])"
python3 -c "import json; d=json.loads(open('$O/bench_2ranks.json').read()); print('2 ranks:', d['n_gpus'], d['value'], d['config'])"
tail -3 $O/bench_2ranks.err
