#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r02p6
mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_ntt.py tests/test_golden.py -x -q -m gpu > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log
rm -f $O/ab.jsonl
for b in 1 0 1 0; do PLONKY2_NTT_XCD=$b PLONKY2_NTT_SKEW=xcd$b python3 tools/scratch/probe5.py >> $O/ab.jsonl 2>&1; done
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python3 $GRAFT_REPO_ROOT/bench.py --no-prove --no-cpu --no-commit --steps 10 --warmup 2 > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT
tail -3 $O/tests.log; cat $O/ab.jsonl
