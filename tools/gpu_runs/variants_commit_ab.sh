#!/bin/bash
# configs[2] commit and the 2^20 transforms: product against the builds in gpurun_in/<name>/ (VARIANTS), interleaved on one device
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/variants_commit_ab; mkdir -p $O; rm -f $O/ab.jsonl
for rep in 1 2; do
  TAG=product python3 tools/gpu_runs/commit_time.py >> $O/ab.jsonl 2>&1
  for v in ${VARIANTS:-old}; do
    TAG=$v PLONKY2_HIP_LIBRARY=$GRAFT_REPO_ROOT/gpurun_in/$v/plonky2_gpu_amd/libplonky2_hip.so python3 tools/gpu_runs/commit_time.py >> $O/ab.jsonl 2>&1
  done
done
cat $O/ab.jsonl
