#!/bin/bash
# LDE / commit path with the direct passes on and off: parity tests, then bit-reversed NTT, coset LDE and commit timings
cd "$GRAFT_REPO_ROOT" || exit 1
DBG=$GRAFT_REPO_ROOT/plonky2_gpu_amd/libplonky2_hip_debug.so
O=gpurun_out/lde_ab; mkdir -p $O; rm -f $O/ab.jsonl
timeout 1500 python3 -m pytest tests/test_gpu_ntt.py tests/test_golden.py tests/test_gpu_merkle.py -x -q -m gpu > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log; tail -n 4 $O/tests.log
for rep in 1 2; do
TAG=direct python3 tools/gpu_runs/lde_time.py >> $O/ab.jsonl 2>&1
TAG=wave_tiles PLONKY2_HIP_LIBRARY=$DBG PLONKY2_NTT_DIRECT=0 python3 tools/gpu_runs/lde_time.py >> $O/ab.jsonl 2>&1
done
cat $O/ab.jsonl
