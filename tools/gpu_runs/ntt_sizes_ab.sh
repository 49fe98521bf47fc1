#!/bin/bash
# NTT parity tests, then timings over sizes with the direct passes on and off
cd "$GRAFT_REPO_ROOT" || exit 1
# the A/B knobs exist in the diagnostic build only (csrc/knobs.h)
DBG=$GRAFT_REPO_ROOT/plonky2_gpu_amd/libplonky2_hip_debug.so
O=gpurun_out/ntt_sizes; mkdir -p $O; rm -f $O/ab.jsonl
timeout 1500 python3 -m pytest tests/test_gpu_ntt.py tests/test_golden.py -x -q -m gpu > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log; tail -n 4 $O/tests.log
for rep in 1 2; do
TAG=direct python3 tools/gpu_runs/ntt_time_sizes.py >> $O/ab.jsonl 2>&1
TAG=wave_tiles PLONKY2_HIP_LIBRARY=$DBG PLONKY2_NTT_DIRECT=0 python3 tools/gpu_runs/ntt_time_sizes.py >> $O/ab.jsonl 2>&1
done
cat $O/ab.jsonl
