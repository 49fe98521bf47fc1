# configs[2] commit (135 columns x 2^20, rate 8, cap height 4) timed alone: bench.py's commit leg, nothing else
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import plonky2_gpu_amd as pg
from plonky2_gpu_amd import _lib
import bench
ctx = pg.Context(0)
r = bench.bench_commit(pg, _lib, ctx, int(os.environ.get("COLS", "135")), int(os.environ.get("LOG_N", "20")), iters=int(os.environ.get("ITERS", "5")))
print(json.dumps({"tag": os.environ.get("TAG"), "commit_ms": round(r["commit_ms"], 3), "without_leaf_major_copy_ms": round(r["commit_ms_without_leaf_major_copy"], 3),
                  "stages_one_at_a_time_ms": r["commit_stage_ms_one_at_a_time"], "cap0": r["cap0"][0]}), flush=True)
