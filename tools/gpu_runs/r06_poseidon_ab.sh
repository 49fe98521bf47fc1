#!/bin/bash
# round 6: the MDS layer fed with the state as it lies (tools/experiments/mds_natural.h; gpurun_in/natural, built by
# tools/gpu_runs/build_variant_files.sh natural '-DPOSEIDON_MDS_NATURAL="../../tools/experiments/mds_natural.h"' merkle fri) against the
# product's byte-plane form: permutations/s and the configs[2] commit, A/B/A/B; VARIANT_TESTS=1 also runs the Merkle parity suite on the variant.
# (The first run of this script had the natural form as the product and the plane form as the variant: profiles/r06_poseidon_natural_layout_ab.jsonl.)
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out; mkdir -p $O
V=$R/gpurun_in/natural/plonky2_gpu_amd/libplonky2_hip.so
[ -n "$VARIANT_TESTS" ] && PLONKY2_HIP_LIBRARY=$V timeout 900 python -m pytest tests/test_gpu_merkle.py -x -q -m gpu 2>&1 | tail -4 | tee $O/r06_poseidon_ab_tests.txt
: > $O/r06_poseidon_ab.jsonl
for rep in 1 2; do
  for v in product natural; do
    if [ $v = natural ]; then export PLONKY2_HIP_LIBRARY=$V; else unset PLONKY2_HIP_LIBRARY; fi
    echo "{\"variant\": \"$v\", \"poseidon\": $(timeout 120 python tools/bench_poseidon.py)}" >> $O/r06_poseidon_ab.jsonl
    TAG=$v ITERS=4 timeout 300 python tools/gpu_runs/commit_time.py >> $O/r06_poseidon_ab.jsonl
  done
done
unset PLONKY2_HIP_LIBRARY
cat $O/r06_poseidon_ab.jsonl
