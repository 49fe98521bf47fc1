#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r02p2
mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_ntt.py tests/test_golden.py tests/test_gpu_merkle.py -x -q -m gpu > $O/tests_wave.log 2>&1
echo "tests rc=$?" >> $O/tests_wave.log
for k in wave tile; do
  PLONKY2_NTT_KERNEL=$k timeout 600 python3 bench.py --no-prove --no-cpu --steps 10 --warmup 2 > $O/bench_$k.json 2> $O/bench_$k.err
done
PLONKY2_NTT_KERNEL=wave timeout 300 python3 tools/ntt_chunk_sweep.py > $O/chunk_wave.jsonl 2>&1
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_wave -- python3 $GRAFT_REPO_ROOT/bench.py --no-prove --no-cpu --no-commit --steps 10 --warmup 2 > $GRAFT_REPO_ROOT/$O/prof_wave.log 2>&1
cd $GRAFT_REPO_ROOT
tail -3 $O/tests_wave.log
for k in wave tile; do python3 -c "import json,sys; d=json.loads(open('$O/bench_$k.json').read()); print('$k', d['value'], d['roofline']['ms'], d['roofline']['frac'], d['extra'].get('commit_ms'))"; done
