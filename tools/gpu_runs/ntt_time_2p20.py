import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from plonky2_gpu_amd import _lib  # PLONKY2_HIP_LIBRARY=<path> selects another build
import plonky2_gpu_amd as pg
ctx = pg.Context(0)
log_n, batch = 20, 64
n = 1 << log_n
rng = np.random.default_rng(1)
host = rng.integers(0, 0xFFFFFFFF00000001, size=(batch, n), dtype=np.uint64)
buf = pg.DeviceBuffer.from_host(ctx, host)
REPS = int(os.environ.get("REPS", "40"))
def t(order, inverse=0):
    ms = []
    for r in range(REPS + 2):
        e0, e1 = pg.Event(), pg.Event()
        e0.record(ctx)
        _lib.call("gl_ntt_batch", buf.ptr, batch, log_n, n, inverse, order, ctx.ptr)
        e1.record(ctx)
        ctx.synchronize()
        if r > 1: ms.append(e1.elapsed_ms_since(e0))
    return round(float(np.median(ms)), 4), round(float(min(ms)), 4)
print(json.dumps({"tag": os.environ.get("TAG"), "wg_per_cu": os.environ.get("PLONKY2_NTT_WG_PER_CU"), "direct": os.environ.get("PLONKY2_NTT_DIRECT"), "chunk": os.environ.get("PLONKY2_NTT_CHUNK_COLS"), "natural_ms": t(0), "inverse_ms": t(0, 1), "bitrev_ms": t(1), "note": "[median, min] over %d launches" % REPS}), flush=True)
