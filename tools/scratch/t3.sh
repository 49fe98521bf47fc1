#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02t3; mkdir -p $O
MASTER_ADDR=127.0.0.1 MASTER_PORT=29701 WORLD_SIZE=1 RANK=0 LOCAL_RANK=0 python3 tests/dist_nccl_self.py > $O/nccl.log 2>&1; echo rc=$? >> $O/nccl.log
grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" $O/nccl.log | tail -12
