#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02p5; mkdir -p $O; rm -f $O/skew.jsonl
for scr in 128 512; do for sk in 0 20 60 120 250; do
  PLONKY2_NTT_SCRATCH_MIB=$scr PLONKY2_NTT_SKEW=$sk python3 tools/scratch/probe5.py >> $O/skew.jsonl 2>&1
done; done
cat $O/skew.jsonl
