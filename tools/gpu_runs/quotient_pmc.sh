#!/bin/bash
# counters of the gate kernels of the ed25519 quotient (tools/bench_quotient_ed25519.py)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/quotient_pmc; rm -rf $O; mkdir -p $O
cd /tmp
pmc() { n=$1; shift; timeout 600 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/pmc_$n -- python3 $R/tools/bench_quotient_ed25519.py 18 3 0 > $O/pmc_$n.log 2>&1; }
pmc sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
pmc sq2 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_IFETCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE SQ_INSTS_VALU
pmc sqc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES
cd $R
python3 - $O <<'PY'
import csv, glob, sys, collections
c = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/pmc_*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "gate_constraints" in r["Kernel_Name"]:
            c[r["Counter_Name"]][int(r["Dispatch_Id"])].append(float(r["Counter_Value"]))
tot = {k: sum(sum(v) for v in d.values()) / max(1, len(d)) * 8 for k, d in c.items()}  # per quotient = 8 unit launches
for k in sorted(tot): print(k, round(tot[k]))
if tot.get("GRBM_GUI_ACTIVE") and tot.get("SQ_INSTS_VALU"):
    cyc = tot["GRBM_GUI_ACTIVE"] / 8
    print("VALU issue frac at 4 cycles:", tot["SQ_INSTS_VALU"] * 4 / (1024 * cyc), "kernel cycles per quotient", cyc)
if tot.get("SQC_ICACHE_REQ"): print("icache hit rate", tot["SQC_ICACHE_HITS"] / tot["SQC_ICACHE_REQ"])
if tot.get("SQ_WAVE_CYCLES"): print("parked", tot["SQ_WAIT_ANY"] / tot["SQ_WAVE_CYCLES"], "issue stall", tot["SQ_WAIT_INST_ANY"] / tot["SQ_WAVE_CYCLES"], "active", tot["SQ_ACTIVE_INST_ANY"] / tot["SQ_WAVE_CYCLES"])
PY
find $O -name "*.csv" -size +6M -delete
