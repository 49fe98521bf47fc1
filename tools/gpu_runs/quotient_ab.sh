#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/quotient_ab; mkdir -p $O; rm -f $O/ab.jsonl
for rep in 1 2; do
for emit in limbs dot; do
  export PLONKY2_HIP_JIT_EMIT=$emit PLONKY2_HIP_KERNEL_CACHE=/tmp/kc_$emit; mkdir -p /tmp/kc_$emit
  timeout 600 python3 tools/bench_quotient_ed25519.py 18 7 0 > $O/$emit.json 2> $O/$emit.err
  echo "$emit $(grep -o '"hiprtc_compile_s": [0-9.]*, "kernel_source_bytes": [0-9]*, "compiled_ms": [0-9.]*' $O/$emit.json)" >> $O/ab.jsonl
done; done
cat $O/ab.jsonl
