#!/bin/bash
# what the memory system gives each of the NTT's access patterns (tools/ubench_mem.hip), built here from source
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/ubench_mem; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_mem.hip -o /tmp/ubench_mem || exit 1
timeout 600 /tmp/ubench_mem > $O/ubench_mem.txt 2>&1; echo rc=$?; grep -n "segs\|contig\|SKIPPED\|fault" $O/ubench_mem.txt | head -60
