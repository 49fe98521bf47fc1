#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/batch_scaling; rm -rf $O; mkdir -p $O
python3 tools/gpu_runs/ntt_batch_scaling.py | tee $O/scaling.json
DATA=zeros python3 tools/gpu_runs/ntt_batch_scaling.py | tee -a $O/scaling.json
cd /tmp
BATCHES=1024 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/tools/gpu_runs/ntt_batch_scaling.py > $O/stats.log 2>&1
cd $R
f=$(find $O/stats -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats_1024.csv
python3 - $O/kernel_stats_1024.csv <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print("  ", r["Name"].split("(")[0].split("::")[-1][:44], r["Calls"], round(float(r["AverageNs"]) / 1e3 / 16, 1), "us per 64 polynomials")
PY
find $O -name "*.csv" -size +6M -delete
