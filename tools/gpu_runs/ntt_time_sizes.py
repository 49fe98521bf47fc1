import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import plonky2_gpu_amd as pg
from plonky2_gpu_amd import _lib
ctx = pg.Context(0)
out = {"tag": os.environ.get("TAG")}
for log_n in (20, 21, 22):
    batch = (1 << 26) >> log_n
    n = 1 << log_n
    rng = np.random.default_rng(1)
    host = rng.integers(0, 0xFFFFFFFF00000001, size=(batch, n), dtype=np.uint64)
    buf = pg.DeviceBuffer.from_host(ctx, host)
    def t(order, inverse=0):
        ms = []
        for r in range(8):
            e0, e1 = pg.Event(), pg.Event()
            e0.record(ctx)
            _lib.call("gl_ntt_batch", buf.ptr, batch, log_n, n, inverse, order, ctx.ptr)
            e1.record(ctx)
            ctx.synchronize()
            if r: ms.append(e1.elapsed_ms_since(e0))
        return float(np.median(ms))
    out[f"2^{log_n}"] = {"natural_ms": round(t(0), 4), "inverse_ms": round(t(0, 1), 4), "bitrev_ms": round(t(1), 4)}
    buf.free()
print(json.dumps(out), flush=True)
