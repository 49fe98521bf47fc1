#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02p3; mkdir -p $O
for w in 2 1; do PLONKY2_NTT_WG_PER_CU=$w python3 tools/scratch/probe3.py > $O/dbg_wg$w.jsonl 2>&1; cat $O/dbg_wg$w.jsonl; done
