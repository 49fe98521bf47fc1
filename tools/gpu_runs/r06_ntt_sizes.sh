#!/bin/bash
# round 6: NTT timings over sizes (product), and 2^22 bit-reversed in two passes in place with the existing kernels (diagnostic build)
cd "$GRAFT_REPO_ROOT" || exit 1
DBG=$GRAFT_REPO_ROOT/plonky2_gpu_amd/libplonky2_hip_debug.so
O=gpurun_out; rm -f $O/r06_ntt_sizes.jsonl
TAG=product python3 tools/gpu_runs/ntt_time_sizes.py >> $O/r06_ntt_sizes.jsonl 2>&1
TAG=three_passes_diagnostic_build SIZES=22 PLONKY2_HIP_LIBRARY=$DBG python3 tools/gpu_runs/ntt_time_sizes.py >> $O/r06_ntt_sizes.jsonl 2>&1
TAG=two_passes_in_place_existing_kernels SIZES=22 PLONKY2_HIP_LIBRARY=$DBG PLONKY2_NTT_TWO_PASS_22_INPLACE=1 python3 tools/gpu_runs/ntt_time_sizes.py >> $O/r06_ntt_sizes.jsonl 2>&1
PLONKY2_HIP_LIBRARY=$DBG PLONKY2_NTT_TWO_PASS_22_INPLACE=1 timeout 600 python3 -m pytest tests/test_gpu_ntt.py -x -q -m gpu -k "bit_rev or bitrev or sizes or 22" 2>&1 | tail -3
cat $O/r06_ntt_sizes.jsonl
