#!/bin/bash
# (a record of a run made BEFORE the generator's switches moved to the diagnostic build, some with knobs of scratch builds that no longer exist:
# to repeat what still applies, export PLONKY2_HIP_LIBRARY=$GRAFT_REPO_ROOT/plonky2_gpu_amd/libplonky2_hip_debug.so)
# round 5: fused units with two-column alpha accumulators (12 registers per gate instead of 18): waves per SIMD x gates per unit x statements ahead, one device
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05fused4; mkdir -p $O
timeout 1200 python3 -m pytest tests/test_reference_quotient.py -x -q -m gpu --durations=3 > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log; tail -n 7 $O/tests.log
: > $O/ab.jsonl
V=$GRAFT_REPO_ROOT/plonky2_gpu_amd/kernel_cache_variants
for rep in 1 2; do
  for spec in "4 4 16" "4 4 32" "4 5 16" "4 5 32" "4 6 16" "4 6 32" "3 6 16" "3 8 16" "4 5 48"; do set -- $spec
    echo "{\"variant\": \"w$1_p$2_d$3\", \"rep\": $rep, \"result\": $(PLONKY2_HIP_JIT_PREFETCH=$3 PLONKY2_HIP_JIT_WAVES=$1 PLONKY2_HIP_JIT_FUSE_GATES=$2 PLONKY2_HIP_KERNEL_CACHE=$V/w$1_p$2_d$3 timeout 400 python3 tools/bench_quotient_ed25519.py 18 7 0 2>/dev/null | tail -n 1)}" >> $O/ab.jsonl
  done
done
python3 - <<'PY'
import json
for l in open("gpurun_out/r05fused4/ab.jsonl"):
    try:
        d = json.loads(l); print(d["variant"], d["rep"], d["result"]["compiled_ms"], d["result"]["hiprtc_compile_s"])
    except Exception as e: print("bad line", l[:100])
PY
