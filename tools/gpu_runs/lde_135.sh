#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/lde135; rm -rf $O; mkdir -p $O
python3 tools/gpu_runs/lde_135.py | tee $O/lde135.json
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/tools/gpu_runs/lde_135.py > $O/stats.log 2>&1
cd $R
f=$(find $O/stats -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats.csv
python3 - $O/kernel_stats.csv <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print("  ", r["Name"].split("(")[0].split("::")[-1][:44], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), "us avg", r["MinNs"], r["MaxNs"])
PY
find $O -name "*.csv" -size +6M -delete
