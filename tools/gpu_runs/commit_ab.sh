#!/bin/bash
# the configs[2] commit with the diagnostic build's knobs
cd "$GRAFT_REPO_ROOT" || exit 1
DBG=$GRAFT_REPO_ROOT/plonky2_gpu_amd/libplonky2_hip_debug.so
O=gpurun_out/commit_ab; mkdir -p $O; rm -f $O/ab.jsonl
for rep in 1 2; do
TAG=product python3 tools/gpu_runs/commit_time.py >> $O/ab.jsonl 2>&1
TAG=debug_build_no_knob PLONKY2_HIP_LIBRARY=$DBG python3 tools/gpu_runs/commit_time.py >> $O/ab.jsonl 2>&1
TAG=one_stage_after_the_other PLONKY2_HIP_LIBRARY=$DBG PLONKY2_COMMIT_PIPELINE=0 python3 tools/gpu_runs/commit_time.py >> $O/ab.jsonl 2>&1
TAG=separate_transposition PLONKY2_HIP_LIBRARY=$DBG PLONKY2_FUSED_LEAVES=0 python3 tools/gpu_runs/commit_time.py >> $O/ab.jsonl 2>&1
TAG=no_direct_ntt PLONKY2_HIP_LIBRARY=$DBG PLONKY2_NTT_DIRECT=0 python3 tools/gpu_runs/commit_time.py >> $O/ab.jsonl 2>&1
done
cat $O/ab.jsonl
