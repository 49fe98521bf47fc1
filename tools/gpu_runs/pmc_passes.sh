#!/bin/bash
# counters + kernel stats of the final kernels -> gpurun_out/${TAG}pmc, summarised into profiles/${TAG}_pmc_summary.json by
# tools/pmc_summary.py (which records the sha256 of the kernel sources: bench.py refuses a summary of other sources)
TAG=${TAG:-r04}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${TAG}pmc
mkdir -p $O
cd /tmp
B="python3 $R/bench.py --no-prove --no-cpu --steps 3 --warmup 1 --inner 1 --windows 0"
# kernel durations of the SAME commands bench.py is run with (steady state: back-to-back launches, default steps / inner / windows):
# the batch transform's two kernels here must add up to the bench line's roofline.ms
timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_bench -- python3 $R/bench.py --no-cpu --no-reference > $O/stats_bench.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_ntt -- python3 $R/bench.py --no-prove --no-cpu --no-commit --no-reference > $O/stats_ntt.log 2>&1
pmc() { n=$1; shift; timeout 900 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/pmc_$n -- $B > $O/pmc_$n.log 2>&1; }
pmc sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
pmc sq2 SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
pmc sq3 SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL GRBM_GUI_ACTIVE
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
pmc tcc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum
# the matrix instructions of the hashing kernels inside the commit (csrc/poseidon.h): bench.py subtracts them from the vector count
pmc mfma SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES
P="python3 $R/tools/bench_poseidon.py"
pp() { n=$1; shift; timeout 600 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/perm_$n -- $P > $O/perm_$n.log 2>&1; }
pp sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
pp sq2 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_IFETCH SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE
# the matrix-core side of the permutation (csrc/poseidon.h): instruction count and the cycles the pipe is busy
pp mfma SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
cd $R
python3 tools/bench_poseidon.py > $O/poseidon_rate.json 2>&1
find $O -name "*.csv" -size +6M -delete
python3 tools/pmc_summary.py $O $TAG | tail -20
cp profiles/${TAG}_pmc_summary.json $O/ 2>/dev/null
du -sh $O
