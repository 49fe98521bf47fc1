#!/bin/bash
# columns per pipeline step of the commit (diagnostic build): prove() at the ed25519 shape and the configs[2] commit
cd "$GRAFT_REPO_ROOT" || exit 1
DBG=$GRAFT_REPO_ROOT/plonky2_gpu_amd/libplonky2_hip_debug.so
O=gpurun_out/chunk_ab; mkdir -p $O; rm -f $O/ab.jsonl
for rep in 1 2; do for ch in 16 32 64 128; do
export PLONKY2_HIP_LIBRARY=$DBG PLONKY2_COMMIT_CHUNK=$ch
timeout 300 python3 tools/bench_prove.py 18 234 5 1 1 2> $O/p.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); b=d['best_ms']
print(json.dumps({'chunk':$ch,'prove_total':b['total'],'wires':b['wires commitment']}))" >> $O/ab.jsonl
TAG=chunk_$ch python3 tools/gpu_runs/commit_time.py >> $O/ab.jsonl 2>&1
done; done
cat $O/ab.jsonl
