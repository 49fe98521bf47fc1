#!/bin/bash
# round 6: the MDS layer with its matrix instructions issued in pairs between independent vector work (tools/experiments/mds_interleave.h,
# gpurun_in/interleave) against the product: parity on the variant, then permutations/s and the configs[2] commit, A/B/A/B
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out; mkdir -p $O
V=$R/gpurun_in/interleave/plonky2_gpu_amd/libplonky2_hip.so
PLONKY2_HIP_LIBRARY=$V timeout 900 python -m pytest tests/test_gpu_merkle.py -x -q -m gpu -k "not full_width and not benchmark_size" 2>&1 | tail -3
: > $O/r06_poseidon_interleave_ab.jsonl
for rep in 1 2; do
  for v in product interleave; do
    if [ $v = interleave ]; then export PLONKY2_HIP_LIBRARY=$V; else unset PLONKY2_HIP_LIBRARY; fi
    echo "{\"variant\": \"$v\", \"poseidon\": $(timeout 120 python tools/bench_poseidon.py)}" >> $O/r06_poseidon_interleave_ab.jsonl
    TAG=$v ITERS=4 timeout 300 python tools/gpu_runs/commit_time.py >> $O/r06_poseidon_interleave_ab.jsonl
  done
done
unset PLONKY2_HIP_LIBRARY
cat $O/r06_poseidon_interleave_ab.jsonl
