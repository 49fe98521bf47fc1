#!/bin/bash
# round 6: the inverse of 2^22 points in two passes (reversed-input column pass): the NTT parity suite, then timings at 2^20..2^23
cd "$GRAFT_REPO_ROOT" || exit 1
DBG=$GRAFT_REPO_ROOT/plonky2_gpu_amd/libplonky2_hip_debug.so
O=gpurun_out
timeout 1500 python3 -m pytest tests/test_gpu_ntt.py tests/test_golden.py -x -q -m gpu 2>&1 | tail -4
rm -f $O/r06_ntt_inverse22.jsonl
TAG=product SIZES=20,21,22,23 python3 tools/gpu_runs/ntt_time_sizes.py >> $O/r06_ntt_inverse22.jsonl 2>&1
TAG=three_passes_diagnostic_build SIZES=22 PLONKY2_HIP_LIBRARY=$DBG PLONKY2_NTT_TWO_PASS_22=0 python3 tools/gpu_runs/ntt_time_sizes.py >> $O/r06_ntt_inverse22.jsonl 2>&1
cat $O/r06_ntt_inverse22.jsonl
