#!/bin/bash
# A/B of the working tree's library against the library built from HEAD~ (gpurun_in/ntt_variants/BASE), 2^20 batch transform
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/ntt_ab; mkdir -p $O; rm -f $O/ab.jsonl
timeout 900 python3 -m pytest tests/test_gpu_ntt.py tests/test_golden.py -x -q -m gpu > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log; tail -n 4 $O/tests.log
V=$GRAFT_REPO_ROOT/gpurun_in/ntt_variants
for rep in 1 2 3; do
TAG=base PLONKY2_HIP_LIBRARY=$V/BASE/libplonky2_hip.so python3 tools/gpu_runs/ntt_time_2p20.py >> $O/ab.jsonl 2>&1
TAG=new python3 tools/gpu_runs/ntt_time_2p20.py >> $O/ab.jsonl 2>&1
done
cat $O/ab.jsonl
