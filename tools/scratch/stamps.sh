#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02stamps; mkdir -p $O
python3 tools/ntt_stamps.py $GRAFT_REPO_ROOT/tools/scratch/stamps/libplonky2_hip.so $O/stamps.jsonl > $O/stamps.txt 2>&1
cat $O/stamps.txt
