#!/bin/bash
# round 5: per-kernel durations of one ed25519 quotient (the gate kernels run slower under the profiler; the others are what is wanted)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05qtrace; rm -rf $O; mkdir -p $O
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- python3 $R/tools/bench_quotient_ed25519.py 18 5 0 > $O/t.log 2>&1
f=$(find $O/t -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats.csv
python3 - $O/kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:24]:
    print(r["Name"][:70].ljust(70), r["Calls"].rjust(6), "avg us %9.1f" % (float(r["AverageNs"]) / 1e3), "total ms %8.2f" % (float(r["TotalDurationNs"]) / 1e6))
PY
find $O -name "*.csv" -size +6M -delete
