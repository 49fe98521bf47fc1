#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/ubench_mem; mkdir -p $O
timeout 600 tools/ubench_mem > $O/ubench_mem.txt 2>&1; echo rc=$?; grep -n "segs\|contig" $O/ubench_mem.txt | head -40
