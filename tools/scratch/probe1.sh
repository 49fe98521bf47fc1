#!/bin/bash
# round-2 probe 1: memory patterns, chunk sweep, SQ/TCC counters of the current kernels
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r02p1
mkdir -p $O
timeout 300 ./tools/ubench_mem > $O/ubench_mem.txt 2>&1
timeout 600 python3 tools/ntt_chunk_sweep.py > $O/chunk_sweep.jsonl 2> $O/chunk_sweep.err
B="python3 bench.py --no-prove --no-cpu --steps 2 --warmup 1"
pmc() { # name, counters...
  n=$1; shift
  timeout 600 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/pmc_$n -- $B > $O/pmc_$n.log 2>&1
}
pmc sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
pmc sq2 SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
pmc sq3 SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE
pmc tcc1 TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum
pmc tcc2 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum
pmc tcp1 TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TA_BUSY_avr
# keep only the CSVs small enough to travel
find $O -name "*.csv" -size +8M -delete
du -sh $O
tail -5 $O/ubench_mem.txt
