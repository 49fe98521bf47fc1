#!/bin/bash
# (a record of a run made BEFORE the generator's switches moved to the diagnostic build, some with knobs of scratch builds that no longer exist:
# to repeat what still applies, export PLONKY2_HIP_LIBRARY=$GRAFT_REPO_ROOT/plonky2_gpu_amd/libplonky2_hip_debug.so)
# A/B of the gate-kernel code generation on one device: wire loads as buffer loads (JITX_LOADS) and ACC as asm multiply-adds (JITX_ACC)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05quotient; mkdir -p $O
: > $O/ab.jsonl
for rep in 1 2; do for v in "0 0" "1 0" "0 1" "1 1"; do set -- $v
  echo "{\"loads\": $1, \"acc\": $2, \"rep\": $rep, \"result\": $(JITX_LOADS=$1 JITX_ACC=$2 timeout 300 python3 tools/bench_quotient_ed25519.py 18 7 0 2>/dev/null | tail -n 1)}" >> $O/ab.jsonl
done; done
python3 - <<'PY'
import json
for l in open("gpurun_out/r05quotient/ab.jsonl"):
    d = json.loads(l); print(d["loads"], d["acc"], d["rep"], d["result"]["compiled_ms"], d["result"]["hiprtc_compile_s"])
PY
