#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02p9; mkdir -p $O; rm -f $O/ab.jsonl
S=$GRAFT_REPO_ROOT/tools/scratch
for rep in 1 2; do
TAG=dual python3 tools/scratch/probe7.py >> $O/ab.jsonl 2>&1
TAG=dual_skeleton PLONKY2_LIB=$S/SKELETON/libplonky2_hip.so python3 tools/scratch/probe7.py >> $O/ab.jsonl 2>&1
TAG=dual_nomem PLONKY2_LIB=$S/NOMEM/libplonky2_hip.so python3 tools/scratch/probe7.py >> $O/ab.jsonl 2>&1
TAG=wave_skeleton PLONKY2_NTT_DUAL=0 PLONKY2_LIB=$S/SKELETON/libplonky2_hip.so python3 tools/scratch/probe7.py >> $O/ab.jsonl 2>&1
done
cat $O/ab.jsonl
