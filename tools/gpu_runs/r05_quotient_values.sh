#!/bin/bash
# round 5: quotient_values_fast_kernel (no scratch, Z_H from a table, one inversion by an addition chain): parity of everything that
# computes quotients, then the ed25519 quotient and the prove stages, then a kernel trace
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05qvalues; rm -rf $O; mkdir -p $O
timeout 2400 python3 -m pytest tests/test_gpu_plonk.py tests/test_reference_quotient.py tests/test_gpu_prove.py tests/test_cpp_prove.py tests/test_reference_dumps.py tests/test_gpu_reference_kernels.py -x -q -m gpu > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log; tail -n 5 $O/tests.log
for rep in 1 2; do timeout 400 python3 tools/bench_quotient_ed25519.py 18 7 0 2>/dev/null | tail -n 1 | grep -o '"compiled_ms": [0-9.]*\|"reference_symbol_ms": [0-9.]*' | tr '\n' ' '; echo; done | tee $O/quotient.txt
timeout 300 python3 tools/bench_prove.py 18 234 5 1 1 > $O/prove.json 2> $O/prove.err; tail -c 900 $O/prove.json
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- python3 $R/tools/bench_quotient_ed25519.py 18 5 0 > $O/t.log 2>&1
f=$(find $O/t -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats.csv
python3 - $O/kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:10]:
    print(r["Name"][:70].ljust(70), r["Calls"].rjust(6), "avg us %9.1f" % (float(r["AverageNs"]) / 1e3), "total ms %8.2f" % (float(r["TotalDurationNs"]) / 1e6))
PY
find $O -name "*.csv" -size +6M -delete
