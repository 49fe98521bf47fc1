#!/bin/bash
# prove() at the ed25519 shape with the diagnostic build's commit knobs
cd "$GRAFT_REPO_ROOT" || exit 1
DBG=$GRAFT_REPO_ROOT/plonky2_gpu_amd/libplonky2_hip_debug.so
O=gpurun_out/prove_ab; mkdir -p $O; rm -f $O/ab.jsonl
run() { tag=$1; shift; env "$@" timeout 300 python3 tools/bench_prove.py 18 234 5 1 1 2> $O/$tag.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); b=d['best_ms']
print(json.dumps({'tag':'$tag','total':b['total'],'wires':b['wires commitment'],'zs':b['zs partial products commitment'],'quotient':b['quotient polys'],'quotient_commit':b['quotient commitment']}))" >> $O/ab.jsonl; }
for rep in 1 2; do
run product X=1
run one_stage_after_the_other PLONKY2_HIP_LIBRARY=$DBG PLONKY2_COMMIT_PIPELINE=0
done
cat $O/ab.jsonl
