#!/bin/bash
# round 6: timeline of proofs at the ed25519 shape — where the device idles (tools/prove_timeline.py)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r06_timeline}
O=$R/gpurun_out/$TAG; rm -rf $O; mkdir -p $O
cd /tmp
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/t -- python3 $R/tools/prove_timeline.py run 18 6 > $O/run.log 2>&1
tail -2 $O/run.log
python3 $R/tools/prove_timeline.py analyse $O/t 6 > $O/timeline.txt 2>&1
cat $O/timeline.txt
find $O -name "*.csv" -size +3M -delete; find $O -name "*.db" -delete
