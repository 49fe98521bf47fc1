#!/bin/bash
# per-kernel durations of the ed25519 quotient (tools/bench_quotient_ed25519.py) with the gates in eight units (product) and in one
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/quotient_stats
rm -rf $O; mkdir -p $O
cd /tmp
for units in 8 1; do
  export PLONKY2_HIP_JIT_UNITS=$units
  [ $units = 1 ] && export PLONKY2_HIP_KERNEL_CACHE=/tmp/kc1 && mkdir -p /tmp/kc1
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/u$units -- python3 $R/tools/bench_quotient_ed25519.py 18 5 0 > $O/u$units.log 2>&1
  f=$(find $O/u$units -name "*kernel_stats.csv" | head -1); cp "$f" $O/units_${units}_kernel_stats.csv
  grep -o '"hiprtc_compile_s": [0-9.]*, "kernel_source_bytes": [0-9]*, "compiled_ms": [0-9.]*' $O/u$units.log
  python3 - "$O/units_${units}_kernel_stats.csv" <<'PY'
import csv, sys
tot = 0
for r in csv.DictReader(open(sys.argv[1])):
    if "gate_constraints" in r["Name"]:
        print("  ", r["Name"][:40], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), "us avg"); tot += float(r["TotalDurationNs"]) / int(r["Calls"])
print("   gate kernels per quotient (sum of averages x launches/quotient): see file")
PY
done
find $O -name "*.csv" -size +6M -delete
