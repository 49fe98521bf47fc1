#!/bin/bash
# round 5: one inversion per thread instead of one per chunk in the partial-products kernel: parity, then the stage with this library
# and one built without the change, alternating, one device
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05pp; mkdir -p $O
timeout 1500 python3 -m pytest tests/test_gpu_plonk.py tests/test_gpu_prove.py tests/test_cpp_prove.py tests/test_reference_dumps.py -x -q -m gpu -k "not full_size" > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log; tail -n 4 $O/tests.log
B=$GRAFT_REPO_ROOT/plonky2_gpu_amd/libplonky2_hip_before_pp.so
: > $O/ab.jsonl
for rep in 1 2 3; do
  echo "{\"lib\": \"before\", \"rep\": $rep, \"prove\": $(PLONKY2_HIP_LIBRARY=$B timeout 300 python3 tools/bench_prove.py 18 234 7 0 1 2>/dev/null | tail -n 1)}" >> $O/ab.jsonl
  echo "{\"lib\": \"after\", \"rep\": $rep, \"prove\": $(timeout 300 python3 tools/bench_prove.py 18 234 7 0 1 2>/dev/null | tail -n 1)}" >> $O/ab.jsonl
done
python3 - <<'PY'
import json
for l in open("gpurun_out/r05pp/ab.jsonl"):
    d = json.loads(l); b = d["prove"]["best_ms"]
    print(d["lib"], d["rep"], "total", b["total"], "partial products", b["partial products"])
PY
