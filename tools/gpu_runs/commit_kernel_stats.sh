#!/bin/bash
# per-kernel durations of the configs[2] commit (tools/gpu_runs/commit_time.py), pipelined (product) and one stage after the other
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/commit_stats
rm -rf $O; mkdir -p $O
cd /tmp
export ITERS=3
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/pipelined -- python3 $R/tools/gpu_runs/commit_time.py > $O/pipelined.log 2>&1
export PLONKY2_HIP_LIBRARY=$R/plonky2_gpu_amd/libplonky2_hip_debug.so PLONKY2_COMMIT_PIPELINE=0
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/serial -- python3 $R/tools/gpu_runs/commit_time.py > $O/serial.log 2>&1
cd $R
for t in pipelined serial; do
  f=$(find $O/$t -name "*kernel_stats.csv" | head -1); cp "$f" $O/${t}_kernel_stats.csv
  tail -n 1 $O/$t.log
  python3 - "$O/${t}_kernel_stats.csv" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:9]:
    print("  ", r["Name"].split("(")[0].split("::")[-1][:40], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), "us avg", round(float(r["TotalDurationNs"]) / 1e6, 2), "ms total")
PY
done
find $O -name "*.csv" -size +6M -delete
