#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02p10; mkdir -p $O; rm -f $O/ab.jsonl
timeout 1200 python3 -m pytest tests/test_gpu_ntt.py tests/test_golden.py tests/test_gpu_merkle.py -x -q -m gpu > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log
tail -5 $O/tests.log
for rep in; do
TAG=split python3 tools/scratch/probe10.py >> $O/ab.jsonl 2>&1
TAG=three_pass PLONKY2_NTT_WIDE=0 python3 tools/scratch/probe10.py >> $O/ab.jsonl 2>&1
done
cat $O/ab.jsonl
