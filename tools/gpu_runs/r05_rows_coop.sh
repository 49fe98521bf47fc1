#!/bin/bash
# round 5: small trees from rows (FRI commit phase): one wavefront per leaf up to 4096 leaves, branch-free s-boxes up to 65 536:
# parity, then prove stages and a small-tree commit with this library and with one built without the change, alternating, one device
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r05rows; mkdir -p $O
timeout 1500 python3 -m pytest tests/test_gpu_merkle.py tests/test_gpu_fri.py tests/test_gpu_prove.py -x -q -m gpu -k "not full_width and not full_size and not benchmark_size" > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log; tail -n 4 $O/tests.log
B=$GRAFT_REPO_ROOT/plonky2_gpu_amd/libplonky2_hip_before_rows.so
: > $O/ab.jsonl
for rep in 1 2 3; do
  echo "{\"lib\": \"before\", \"rep\": $rep, \"prove\": $(PLONKY2_HIP_LIBRARY=$B timeout 300 python3 tools/bench_prove.py 18 234 7 0 1 2>/dev/null | tail -n 1)}" >> $O/ab.jsonl
  echo "{\"lib\": \"after\", \"rep\": $rep, \"prove\": $(timeout 300 python3 tools/bench_prove.py 18 234 7 0 1 2>/dev/null | tail -n 1)}" >> $O/ab.jsonl
done
python3 - <<'PY'
import json
for l in open("gpurun_out/r05rows/ab.jsonl"):
    d = json.loads(l); b = d["prove"]["best_ms"]
    print(d["lib"], d["rep"], "total", b["total"], "without syncs", d["prove"]["best_ms_without_stage_syncs"], "fri commit", b["fri: commit phase"], "combine", b["fri: combine + divide"],
          "zs commit", b["zs partial products commitment"], "quotient commit", b["quotient commitment"], "wires", b["wires commitment"])
PY
