#!/bin/bash
# round 6: kernel trace of two proofs in flight (two host threads, own contexts): which kernels of one proof run beside which of the other
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06_inflight_trace; rm -rf $O; mkdir -p $O
cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/t -- python3 $R/tools/prove_timeline.py run_in_flight 18 6 2 > $O/run.log 2>&1
tail -1 $O/run.log
python3 $R/tools/prove_timeline.py overlap $O/t 12 > $O/overlap.txt 2>&1
cat $O/overlap.txt
find $O -name "*.csv" -size +3M -delete; find $O -name "*.db" -delete
