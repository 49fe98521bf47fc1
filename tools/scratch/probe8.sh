#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02p8; mkdir -p $O; rm -f $O/ab.jsonl
timeout 900 python3 -m pytest tests/test_gpu_ntt.py tests/test_golden.py -x -q -m gpu > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log
tail -5 $O/tests.log
for rep in 1 2; do
TAG=dual python3 tools/scratch/probe7.py >> $O/ab.jsonl 2>&1
TAG=wave PLONKY2_NTT_DUAL=0 python3 tools/scratch/probe7.py >> $O/ab.jsonl 2>&1
done
cat $O/ab.jsonl
