#!/bin/bash
# round 6: 2^22 bit-reversed in two passes in place with the two-waves-per-row kernel for 2048-point rows: parity, then timings
# (product = the new plan; diagnostic build: the three-pass plan, and the same two passes with the round-1 kernel over the rows)
cd "$GRAFT_REPO_ROOT" || exit 1
DBG=$GRAFT_REPO_ROOT/plonky2_gpu_amd/libplonky2_hip_debug.so
O=gpurun_out
timeout 1500 python3 -m pytest tests/test_gpu_ntt.py tests/test_golden.py -x -q -m gpu 2>&1 | tail -4
rm -f $O/r06_ntt_bitrev22.jsonl
TAG=product SIZES=20,21,22 python3 tools/gpu_runs/ntt_time_sizes.py >> $O/r06_ntt_bitrev22.jsonl 2>&1
TAG=three_passes_diagnostic_build SIZES=22 PLONKY2_HIP_LIBRARY=$DBG PLONKY2_NTT_TWO_PASS_22_INPLACE=0 python3 tools/gpu_runs/ntt_time_sizes.py >> $O/r06_ntt_bitrev22.jsonl 2>&1
TAG=two_passes_round1_row_kernel SIZES=22 PLONKY2_HIP_LIBRARY=$DBG PLONKY2_NTT_TWO_PASS_22_INPLACE=generic python3 tools/gpu_runs/ntt_time_sizes.py >> $O/r06_ntt_bitrev22.jsonl 2>&1
cat $O/r06_ntt_bitrev22.jsonl
