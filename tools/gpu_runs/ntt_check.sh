#!/bin/bash
# NTT parity (tests + a large-size differential campaign) and timings over sizes, with and without the sixteen-wave tiles
cd "$GRAFT_REPO_ROOT" || exit 1
# the A/B knobs exist in the diagnostic build only (csrc/knobs.h)
DBG=$GRAFT_REPO_ROOT/plonky2_gpu_amd/libplonky2_hip_debug.so
O=gpurun_out/ntt_check; mkdir -p $O; rm -f $O/ab.jsonl
timeout 1800 python3 -m pytest tests/test_gpu_ntt.py tests/test_golden.py -x -q -m gpu > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log
tail -n 5 $O/tests.log
timeout 1500 python3 tests/fuzz_commit.py 40 ${1:-7} large > $O/fuzz_large.log 2>&1; echo "fuzz rc=$?"; tail -n 1 $O/fuzz_large.log
for rep in 1 2; do
TAG=product python3 tools/gpu_runs/ntt_time_sizes.py >> $O/ab.jsonl 2>&1
TAG=narrow_tiles PLONKY2_HIP_LIBRARY=$DBG PLONKY2_NTT_WIDE=0 python3 tools/gpu_runs/ntt_time_sizes.py >> $O/ab.jsonl 2>&1
done
cat $O/ab.jsonl
[ -x tools/ubench_issue ] && tools/ubench_issue > $O/ubench_issue.txt 2>&1 && head -n 28 $O/ubench_issue.txt
