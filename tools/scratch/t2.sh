#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02t2; mkdir -p $O
timeout 2400 python3 -m pytest tests/test_reference_dumps.py -x -q -m gpu > $O/t2.log 2>&1
echo "rc=$?" >> $O/t2.log
tail -40 $O/t2.log
