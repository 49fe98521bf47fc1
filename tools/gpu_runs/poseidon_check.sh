#!/bin/bash
# Poseidon / Merkle / FRI / proof parity, then the permutation rate (product and vector-ALU implementation) and the configs[2] commit
cd "$GRAFT_REPO_ROOT" || exit 1
DBG=$GRAFT_REPO_ROOT/plonky2_gpu_amd/libplonky2_hip_debug.so
O=gpurun_out/poseidon; mkdir -p $O; rm -f $O/rates.jsonl
timeout 2400 python3 -m pytest tests/test_gpu_merkle.py tests/test_golden.py tests/test_gpu_fri.py tests/test_gpu_prove.py -x -q -m gpu > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log; tail -n 5 $O/tests.log
for rep in 1 2; do
python3 tools/bench_poseidon.py >> $O/rates.jsonl 2>&1
PLONKY2_HIP_LIBRARY=$DBG PLONKY2_POSEIDON=vector python3 tools/bench_poseidon.py >> $O/rates.jsonl 2>&1
TAG=product python3 tools/gpu_runs/commit_time.py >> $O/rates.jsonl 2>&1
done
cat $O/rates.jsonl
