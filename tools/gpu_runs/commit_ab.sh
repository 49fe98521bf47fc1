#!/bin/bash
# commit parity tests, then the configs[2] commit with the diagnostic build's knobs
cd "$GRAFT_REPO_ROOT" || exit 1
DBG=$GRAFT_REPO_ROOT/plonky2_gpu_amd/libplonky2_hip_debug.so
O=gpurun_out/commit_ab; mkdir -p $O; rm -f $O/ab.jsonl
timeout 1500 python3 -m pytest tests/test_gpu_merkle.py tests/test_golden.py -x -q -m gpu > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log; tail -n 4 $O/tests.log
for rep in 1 2; do
TAG=product python3 tools/gpu_runs/commit_time.py >> $O/ab.jsonl 2>&1
TAG=separate_transposition PLONKY2_HIP_LIBRARY=$DBG PLONKY2_FUSED_LEAVES=0 python3 tools/gpu_runs/commit_time.py >> $O/ab.jsonl 2>&1
TAG=no_direct_ntt PLONKY2_HIP_LIBRARY=$DBG PLONKY2_NTT_DIRECT=0 python3 tools/gpu_runs/commit_time.py >> $O/ab.jsonl 2>&1
done
cat $O/ab.jsonl
