#!/bin/bash
# (a record of a run made BEFORE the generator's switches moved to the diagnostic build, some with knobs of scratch builds that no longer exist:
# to repeat what still applies, export PLONKY2_HIP_LIBRARY=$GRAFT_REPO_ROOT/plonky2_gpu_amd/libplonky2_hip_debug.so)
# round 5: fused units of the gate-kernel generator: parity of everything that runs compiled gates, then the ed25519 quotient for
# waves per SIMD x gates per unit (code objects precompiled into plonky2_gpu_amd/kernel_cache_variants/), one device
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05fused; mkdir -p $O
timeout 2400 python3 -m pytest tests/test_reference_quotient.py tests/test_gpu_plonk.py tests/test_gpu_prove.py tests/test_cpp_prove.py tests/test_reference_dumps.py tests/test_gpu_reference_kernels.py -x -q -m gpu --durations=8 > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log; tail -n 14 $O/tests.log
: > $O/ab.jsonl
V=$GRAFT_REPO_ROOT/plonky2_gpu_amd/kernel_cache_variants
for rep in 1 2; do
  echo "{\"variant\": \"nofuse\", \"rep\": $rep, \"result\": $(PLONKY2_HIP_JIT_FUSE=0 PLONKY2_HIP_KERNEL_CACHE=$V/nofuse timeout 400 python3 tools/bench_quotient_ed25519.py 18 7 0 2>/dev/null | tail -n 1)}" >> $O/ab.jsonl
  for spec in "4 3" "4 4" "4 5" "3 4" "3 5" "3 6" "2 6" "2 8" "2 12"; do set -- $spec
    echo "{\"variant\": \"w$1_p$2\", \"rep\": $rep, \"result\": $(PLONKY2_HIP_JIT_WAVES=$1 PLONKY2_HIP_JIT_FUSE_GATES=$2 PLONKY2_HIP_KERNEL_CACHE=$V/w$1_p$2 timeout 400 python3 tools/bench_quotient_ed25519.py 18 7 0 2>/dev/null | tail -n 1)}" >> $O/ab.jsonl
  done
done
python3 - <<'PY'
import json
for l in open("gpurun_out/r05fused/ab.jsonl"):
    try:
        d = json.loads(l); print(d["variant"], d["rep"], d["result"]["compiled_ms"], d["result"]["hiprtc_compile_s"])
    except Exception as e: print("bad line", l[:100])
PY
timeout 300 python3 tools/bench_prove.py 18 234 5 1 1 > $O/prove.json 2> $O/prove.err; tail -c 1200 $O/prove.json
