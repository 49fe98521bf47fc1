#!/bin/bash
# Counters of the NTT passes alone (2^20, 64 columns: natural forward, inverse, bit-reversed) -> gpurun_out/nttpmc
# Summarise with: python tools/pmc_summary.py gpurun_out/nttpmc <tag>
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/nttpmc
rm -rf $O; mkdir -p $O
cd /tmp
export REPS=10
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_ntt -- python3 $R/tools/gpu_runs/ntt_time_2p20.py > $O/stats_ntt.log 2>&1
pmc() { n=$1; shift; timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/pmc_$n -- python3 $R/tools/gpu_runs/ntt_time_2p20.py > $O/pmc_$n.log 2>&1; }
pmc sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
pmc sq2 SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
pmc sq3 SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL GRBM_GUI_ACTIVE
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
cd $R
find $O -name "*.csv" -size +6M -delete
python3 tools/pmc_summary.py $O tmp_ntt 2>&1 | tail -12
cp profiles/tmp_ntt_pmc_summary.json $O/summary.json; rm -f profiles/tmp_ntt_pmc_summary.json
