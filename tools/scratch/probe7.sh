#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02p7; mkdir -p $O; rm -f $O/ab.jsonl
for rep in 1 2; do
TAG=product python3 tools/scratch/probe7.py >> $O/ab.jsonl 2>&1
TAG=skeleton PLONKY2_LIB=$GRAFT_REPO_ROOT/tools/scratch/skel/libplonky2_hip.so python3 tools/scratch/probe7.py >> $O/ab.jsonl 2>&1
TAG=skeleton PLONKY2_NTT_WG_PER_CU=1 PLONKY2_LIB=$GRAFT_REPO_ROOT/tools/scratch/skel/libplonky2_hip.so python3 tools/scratch/probe7.py >> $O/ab.jsonl 2>&1
TAG=product PLONKY2_NTT_WG_PER_CU=1 python3 tools/scratch/probe7.py >> $O/ab.jsonl 2>&1
done
cat $O/ab.jsonl
