#!/bin/bash
# 2^20 batch transform by the number of columns per launch pair (natural order goes through the workspace in chunks)
cd "$GRAFT_REPO_ROOT" || exit 1
# the A/B knobs exist in the diagnostic build only (csrc/knobs.h)
DBG=$GRAFT_REPO_ROOT/plonky2_gpu_amd/libplonky2_hip_debug.so
O=gpurun_out/ntt_chunks; mkdir -p $O; rm -f $O/ab.jsonl
for rep in 1 2; do
for c in 16 32 64 8; do
TAG=new PLONKY2_HIP_LIBRARY=$DBG PLONKY2_NTT_CHUNK_COLS=$c python3 tools/gpu_runs/ntt_time_2p20.py >> $O/ab.jsonl 2>&1
done
done
cat $O/ab.jsonl
