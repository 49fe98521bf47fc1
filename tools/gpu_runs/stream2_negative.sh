#!/bin/bash
# tests/test_gpu_stream2.py must PASS on the product and on the diagnostic build, and FAIL on the diagnostic build with
# PLONKY2_DROP_STREAM2_WAIT=1 (the order behind the caller's stream2 work removed): shows that the test notices that loss.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/stream2_negative; mkdir -p $O
DBG=$GRAFT_REPO_ROOT/plonky2_gpu_amd/libplonky2_hip_debug.so
timeout 600 python3 -m pytest tests/test_gpu_stream2.py -m gpu -q > $O/product.log 2>&1; echo "product rc=$?" | tee $O/summary.txt
PLONKY2_HIP_LIBRARY=$DBG timeout 600 python3 -m pytest tests/test_gpu_stream2.py -m gpu -q > $O/debug.log 2>&1; echo "diagnostic build rc=$?" | tee -a $O/summary.txt
PLONKY2_HIP_LIBRARY=$DBG PLONKY2_DROP_STREAM2_WAIT=1 timeout 600 python3 -m pytest tests/test_gpu_stream2.py -m gpu -q > $O/dropped.log 2>&1
echo "diagnostic build, waits dropped rc=$? (must be non-zero)" | tee -a $O/summary.txt
tail -5 $O/product.log $O/debug.log; grep -E "^FAILED|passed|failed" $O/dropped.log | tail -8 | tee -a $O/summary.txt
