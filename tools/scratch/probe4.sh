#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r02p4
mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_ntt.py tests/test_golden.py tests/test_gpu_merkle.py -x -q -m gpu > $O/tests_wave.log 2>&1
echo "tests rc=$?" >> $O/tests_wave.log
for w in 1 0; do
  PLONKY2_NTT_WIDE=$w timeout 600 python3 bench.py --no-prove --no-cpu --no-commit --steps 10 --warmup 2 > $O/bench_wide$w.json 2> $O/bench_wide$w.err
  PLONKY2_NTT_WIDE=$w timeout 300 python3 tools/ntt_chunk_sweep.py > $O/chunk_wide$w.jsonl 2>&1
done
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python3 $GRAFT_REPO_ROOT/bench.py --no-prove --no-cpu --no-commit --steps 10 --warmup 2 > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT
tail -3 $O/tests_wave.log
for k in wide1 wide0; do python3 -c "import json,sys; d=json.loads(open('$O/bench_$k.json').read()); print('$k', d['value'], d['roofline']['ms'], d['roofline']['frac'])"; done
grep -h -E '"chunk_cols": (8|16|32|64)' $O/chunk_wide1.jsonl
