#!/bin/bash
# NTT parity, then the headline leg of bench.py alone (plain and under the kernel trace)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/ntt_quick; rm -rf $O; mkdir -p $O
timeout 1800 python3 -m pytest tests/test_gpu_ntt.py tests/test_golden.py -x -q -m gpu > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log; tail -n 3 $O/tests.log
for rep in 1 2; do
python3 bench.py --no-prove --no-cpu --no-commit 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), d['roofline']['ms'], round(d['roofline']['frac'],4))"
done
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --no-prove --no-cpu --no-commit --steps 20 --warmup 3 --inner 1 --windows 0 > $O/stats.log 2>&1
cd $R
f=$(find $O/stats -name "*kernel_stats.csv" | head -1); cp "$f" $O/kernel_stats.csv
python3 - $O/kernel_stats.csv <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:5]:
    print("  ", r["Name"].split("(")[0].split("::")[-1][:50], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), "us avg")
PY
find $O -name "*.csv" -size +6M -delete
