#!/bin/bash
# per-kernel durations of the bit-reversed NTT and the coset LDE (tools/gpu_runs/lde_time.py) -> gpurun_out/lde_stats
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/lde_stats
rm -rf $O; mkdir -p $O
cd /tmp
export SIZES=${SIZES:-20}
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/tools/gpu_runs/lde_time.py > $O/stats.log 2>&1
cd $R
f=$(find $O/stats -name "*kernel_stats.csv" | head -1)
cp "$f" $O/kernel_stats.csv
find $O -name "*.csv" -size +6M -delete
cut -c1-200 $O/kernel_stats.csv | head -12
