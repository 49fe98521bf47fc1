#!/bin/bash
# A/B of library builds in gpurun_in/<name>/plonky2_gpu_amd/libplonky2_hip.so against the product, interleaved, same device
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/ntt_variants_ab; mkdir -p $O; rm -f $O/ab.jsonl
for rep in 1 2 3; do
  TAG=product python3 tools/gpu_runs/ntt_time_2p20.py >> $O/ab.jsonl 2>&1
  for v in ${VARIANTS:-base}; do
    TAG=$v PLONKY2_HIP_LIBRARY=$GRAFT_REPO_ROOT/gpurun_in/$v/plonky2_gpu_amd/libplonky2_hip.so python3 tools/gpu_runs/ntt_time_2p20.py >> $O/ab.jsonl 2>&1
  done
done
python3 - <<PY
import json, collections
d = collections.defaultdict(lambda: collections.defaultdict(list))
for l in open("$O/ab.jsonl"):
    try: r = json.loads(l)
    except Exception: print(l[:200]); continue
    for k in ("natural_ms", "inverse_ms", "bitrev_ms"): d[r["tag"]][k].append(r[k][0])
for t, v in d.items(): print(t, {k: round(sorted(x)[len(x)//2], 4) for k, x in v.items()})
PY
