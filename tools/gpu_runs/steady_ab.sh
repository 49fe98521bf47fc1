#!/bin/bash
# steady-state 2^20 batch timings (tools/gpu_runs/ntt_time_sizes.py): product against the builds in gpurun_in/<name>/ (VARIANTS), interleaved
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/steady_ab; mkdir -p $O; rm -f $O/ab.jsonl
for rep in 1 2 3; do
  TAG=product SIZES=${SIZES:-20} python3 tools/gpu_runs/ntt_time_sizes.py >> $O/ab.jsonl 2>&1 < /dev/null
  for v in ${VARIANTS:-prev}; do
    TAG=$v SIZES=${SIZES:-20} PLONKY2_HIP_LIBRARY=$GRAFT_REPO_ROOT/gpurun_in/$v/plonky2_gpu_amd/libplonky2_hip.so python3 tools/gpu_runs/ntt_time_sizes.py >> $O/ab.jsonl 2>&1 < /dev/null
  done
done
python3 - <<PY
import json, collections
d = collections.defaultdict(lambda: collections.defaultdict(list))
for l in open("$O/ab.jsonl"):
    try: r = json.loads(l)
    except Exception: print(l[:200]); continue
    for sz, v in r.items():
        if sz.startswith("2^"):
            for k in ("natural_ms", "inverse_ms", "bitrev_ms"): d[r["tag"]][sz + " " + k].append(v[k])
for t, v in d.items(): print(t, {k: sorted(x) for k, x in v.items()})
PY
