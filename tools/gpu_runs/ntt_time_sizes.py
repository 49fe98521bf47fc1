import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import plonky2_gpu_amd as pg
from plonky2_gpu_amd import _lib
ctx = pg.Context(0)
out = {"tag": os.environ.get("TAG"), "direct": os.environ.get("PLONKY2_NTT_DIRECT"), "note": "512 MiB batches; ms per batch, median of 12"}
for log_n in [int(x) for x in os.environ.get("SIZES", "16,18,20,21,22,23").split(",")]:
    batch = (1 << 26) >> log_n
    n = 1 << log_n
    rng = np.random.default_rng(1)
    host = rng.integers(0, 0xFFFFFFFF00000001, size=(batch, n), dtype=np.uint64)
    buf = pg.DeviceBuffer.from_host(ctx, host)
    def t(order, inverse=0):
        ms = []
        for r in range(14):
            e0, e1 = pg.Event(), pg.Event()
            e0.record(ctx)
            _lib.call("gl_ntt_batch", buf.ptr, batch, log_n, n, inverse, order, ctx.ptr)
            e1.record(ctx)
            ctx.synchronize()
            if r > 1: ms.append(e1.elapsed_ms_since(e0))
        return float(np.median(ms))
    r = {"natural_ms": round(t(0), 4), "inverse_ms": round(t(0, 1), 4), "bitrev_ms": round(t(1), 4)}
    r["natural_frac_of_8TBps"] = round(16.0 * batch * n / (r["natural_ms"] * 1e-3) / 8e12, 4)
    out[f"2^{log_n}"] = r
    buf.free()
print(json.dumps(out), flush=True)
