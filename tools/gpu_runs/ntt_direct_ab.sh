#!/bin/bash
# Direct passes against the wave-tile kernels: parity tests, then the 2^20 batch transform timed with the base build
# (gpurun_in/ntt_variants/BASE = library of HEAD), with the working tree's library and direct passes off, and with them on.
cd "$GRAFT_REPO_ROOT" || exit 1
# the A/B knobs exist in the diagnostic build only (csrc/knobs.h)
DBG=$GRAFT_REPO_ROOT/plonky2_gpu_amd/libplonky2_hip_debug.so
O=gpurun_out/ntt_direct; mkdir -p $O; rm -f $O/ab.jsonl
timeout 1200 python3 -m pytest tests/test_gpu_ntt.py tests/test_golden.py -x -q -m gpu > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log; tail -n 6 $O/tests.log
V=$GRAFT_REPO_ROOT/gpurun_in/ntt_variants
for rep in 1 2 3; do
[ -f $V/BASE/libplonky2_hip.so ] && TAG=base PLONKY2_HIP_LIBRARY=$V/BASE/libplonky2_hip.so python3 tools/gpu_runs/ntt_time_2p20.py >> $O/ab.jsonl 2>&1
TAG=new PLONKY2_HIP_LIBRARY=$DBG PLONKY2_NTT_DIRECT=0 python3 tools/gpu_runs/ntt_time_2p20.py >> $O/ab.jsonl 2>&1
TAG=new PLONKY2_HIP_LIBRARY=$DBG PLONKY2_NTT_DIRECT=1 python3 tools/gpu_runs/ntt_time_2p20.py >> $O/ab.jsonl 2>&1
done
cat $O/ab.jsonl
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python3 $GRAFT_REPO_ROOT/tools/gpu_runs/ntt_time_2p20.py > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT; find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -r head -n 12
