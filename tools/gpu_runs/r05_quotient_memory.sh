#!/bin/bash
# (a record of a run made BEFORE the generator's switches moved to the diagnostic build, some with knobs of scratch builds that no longer exist:
# to repeat what still applies, export PLONKY2_HIP_LIBRARY=$GRAFT_REPO_ROOT/plonky2_gpu_amd/libplonky2_hip_debug.so)
# round 5: is the ed25519 quotient's gate kernel bound by its loads? (1) HBM-side and L2 counters of the unit kernels, (2) the same
# arithmetic with every wire load redirected to 4 columns (experiment knob, wrong results), (3) units = 1 / 2 / 4 / 8
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05qmem; rm -rf $O; mkdir -p $O
echo "baseline:  $(timeout 400 python3 tools/bench_quotient_ed25519.py 18 7 0 2>/dev/null | tail -n 1 | grep -o '"compiled_ms": [0-9.]*')" | tee -a $O/summary.txt
for k in 4 64; do
echo "samewire $k: $(JITX_SAMEWIRE=$k PLONKY2_HIP_KERNEL_CACHE=/tmp/kc_same$k PLONKY2_HIP_JIT_FORK=1 timeout 600 python3 tools/bench_quotient_ed25519.py 18 7 0 2>/dev/null | tail -n 1 | grep -o '"hiprtc_compile_s": [0-9.]*\|"compiled_ms": [0-9.]*' | tr '\n' ' ')" | tee -a $O/summary.txt
done
for u in 1 2 4; do
mkdir -p /tmp/kcu$u
echo "units $u: $(PLONKY2_HIP_JIT_UNITS=$u PLONKY2_HIP_KERNEL_CACHE=/tmp/kcu$u timeout 900 python3 tools/bench_quotient_ed25519.py 18 7 0 2>/dev/null | tail -n 1 | grep -o '"hiprtc_compile_s": [0-9.]*\|"compiled_ms": [0-9.]*' | tr '\n' ' ')" | tee -a $O/summary.txt
done
cd /tmp
pmc() { n=$1; shift; timeout 600 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/pmc_$n -- python3 $R/tools/bench_quotient_ed25519.py 18 2 0 > $O/pmc_$n.log 2>&1; }
pmc fetch FETCH_SIZE TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
pmc write WRITE_SIZE TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum TCP_TCC_READ_REQ_sum
pmc tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE
cd $R
python3 - $O <<'PY' | tee -a $O/summary.txt
import csv, glob, sys, collections
c = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(set)
for f in glob.glob(sys.argv[1] + "/pmc_*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "gate_constraints" in r["Kernel_Name"]:
            c[r["Counter_Name"]][int(r["Dispatch_Id"])] += float(r["Counter_Value"])
for k in sorted(c):
    d = c[k]; per_launch = sum(d.values()) / len(d)
    print(k, "per unit launch", round(per_launch), "per quotient (x8)", round(per_launch * 8), "launches seen", len(d))
PY
find $O -name "*.csv" -size +6M -delete
