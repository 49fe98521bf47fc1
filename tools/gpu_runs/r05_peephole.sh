#!/bin/bash
# (a record of a run made BEFORE the generator's switches moved to the diagnostic build, some with knobs of scratch builds that no longer exist:
# to repeat what still applies, export PLONKY2_HIP_LIBRARY=$GRAFT_REPO_ROOT/plonky2_gpu_amd/libplonky2_hip_debug.so)
# round 5: the peephole pass of the gate-kernel generator (short forms for small constants, a limb's range check as a square): parity of
# everything that runs compiled gates, then the ed25519 quotient with the pass and without it on one device, then the prove stages
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05peephole; mkdir -p $O
timeout 2400 python3 -m pytest tests/test_reference_quotient.py tests/test_gpu_plonk.py tests/test_gpu_prove.py tests/test_cpp_prove.py tests/test_reference_dumps.py tests/test_gpu_reference_kernels.py -x -q -m gpu --durations=8 > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log; tail -n 14 $O/tests.log
: > $O/ab.jsonl
for rep in 1 2; do for v in 0 1; do
  echo "{\"peephole\": $v, \"rep\": $rep, \"result\": $(PLONKY2_HIP_JIT_PEEPHOLE=$v timeout 400 python3 tools/bench_quotient_ed25519.py 18 7 0 2>/dev/null | tail -n 1)}" >> $O/ab.jsonl
done; done
python3 - <<'PY'
import json
for l in open("gpurun_out/r05peephole/ab.jsonl"):
    d = json.loads(l); print(d["peephole"], d["rep"], d["result"]["compiled_ms"], d["result"]["hiprtc_compile_s"])
PY
timeout 300 python3 tools/bench_prove.py 18 234 5 1 1 > $O/prove.json 2> $O/prove.err; tail -c 1200 $O/prove.json
