#!/bin/bash
# round 5, third run: the two-pass plan for 2^22 (parity, then A/B against the three-pass plan on the same device), the full-width sweep
mkdir -p gpurun_out
python3 -m pytest tests/test_gpu_ntt.py -x -q -m gpu > gpurun_out/r05_ntt_tests.log 2>&1
echo "rc=$?" >> gpurun_out/r05_ntt_tests.log
D=$PWD/plonky2_gpu_amd/libplonky2_hip_debug.so
: > gpurun_out/r05_ntt_sizes.jsonl
for rep in 1 2; do
  TAG="r05 two-pass 2^22 (product)" SIZES=20,21,22,23 python3 tools/gpu_runs/ntt_time_sizes.py >> gpurun_out/r05_ntt_sizes.jsonl 2>&1
  TAG="r05 three-pass 2^22 (diagnostic build, PLONKY2_NTT_TWO_PASS_22=0)" SIZES=22 PLONKY2_HIP_LIBRARY=$D PLONKY2_NTT_TWO_PASS_22=0 python3 tools/gpu_runs/ntt_time_sizes.py >> gpurun_out/r05_ntt_sizes.jsonl 2>&1
done
python3 tools/sweep.py > gpurun_out/r05_sweep.jsonl 2> gpurun_out/r05_sweep.err
tail -n 3 gpurun_out/r05_ntt_tests.log; cat gpurun_out/r05_ntt_sizes.jsonl | cut -c1-700; tail -n 8 gpurun_out/r05_sweep.jsonl; tail -n 3 gpurun_out/r05_sweep.err
