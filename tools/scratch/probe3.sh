#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02p3; mkdir -p $O
python3 tools/scratch/probe3.py > $O/dbg2.jsonl 2>&1; cat $O/dbg2.jsonl
