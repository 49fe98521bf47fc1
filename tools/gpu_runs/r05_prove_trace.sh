#!/bin/bash
# round 5: kernel trace of whole proofs at the ed25519 shape (launch counts and the small kernels of the FRI / opening stages)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05provetrace; rm -rf $O; mkdir -p $O
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- python3 $R/tools/bench_prove.py 18 234 5 1 1 > $O/t.log 2>&1
f=$(find $O/t -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats.csv
python3 - $O/kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot_calls = sum(int(r["Calls"]) for r in rows)
print("kernels launched in the run:", tot_calls)
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:45]:
    print(r["Name"][:78].ljust(78), r["Calls"].rjust(6), "avg us %9.1f" % (float(r["AverageNs"]) / 1e3), "total ms %8.2f" % (float(r["TotalDurationNs"]) / 1e6))
PY
find $O -name "*.csv" -size +6M -delete
