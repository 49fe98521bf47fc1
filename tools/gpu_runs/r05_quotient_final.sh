#!/bin/bash
# round 5: the ed25519 quotient with the final generator: timing beside the interpreter and the reference symbol, then FETCH_SIZE and
# L2 counters of the fused unit kernels (to set against profiles/r05_quotient_loads_experiment.txt)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05qfinal; rm -rf $O; mkdir -p $O
timeout 600 python3 tools/bench_quotient_ed25519.py 18 7 1 > $O/quotient.json 2> $O/quotient.err; tail -c 1200 $O/quotient.json
cd /tmp
pmc() { n=$1; shift; timeout 600 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/pmc_$n -- python3 $R/tools/bench_quotient_ed25519.py 18 2 0 > $O/pmc_$n.log 2>&1; }
pmc fetch FETCH_SIZE
pmc tcc TCC_HIT_sum TCC_MISS_sum
pmc sq SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
cd $R
python3 - $O <<'PY' | tee $O/summary.txt
import csv, glob, sys, collections
c = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(sys.argv[1] + "/pmc_*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "gate_constraints" in r["Kernel_Name"]:
            c[r["Counter_Name"]][int(r["Dispatch_Id"])] += float(r["Counter_Value"])
launches = None
for k in sorted(c):
    d = c[k]
    # the units of one quotient are consecutive dispatches; the bench runs 3 quotients (1 warm-up + 2): sum / 3
    print(k, "per quotient", round(sum(d.values()) / 3), "gate-kernel dispatches seen", len(d))
PY
find $O -name "*.csv" -size +6M -delete
