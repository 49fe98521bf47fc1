import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import plonky2_gpu_amd as pg
from plonky2_gpu_amd import _lib
ctx = pg.Context(0)
out = {"tag": os.environ.get("TAG"), "direct": os.environ.get("PLONKY2_NTT_DIRECT"), "note": "512 MiB batches; ms per batch in steady state: 0.15 s of launches first, then the median of five groups of ~40 ms of launches back to back"}
for log_n in [int(x) for x in os.environ.get("SIZES", "16,18,19,20,21,22,23,24").split(",")]:
    batch = (1 << 26) >> log_n
    n = 1 << log_n
    rng = np.random.default_rng(1)
    host = rng.integers(0, 0xFFFFFFFF00000001, size=(batch, n), dtype=np.uint64)
    buf = pg.DeviceBuffer.from_host(ctx, host)
    def t(order, inverse=0):
        # steady state, as bench.py measures the headline: ~0.15 s of launches first (after an upload the clocks are down and the
        # first dozen launches run 8-10 % slower), then five groups of launches back to back between one pair of events each
        def group(k):
            e0, e1 = pg.Event(), pg.Event()
            e0.record(ctx)
            for _ in range(k):
                _lib.call("gl_ntt_batch", buf.ptr, batch, log_n, n, inverse, order, ctx.ptr)
            e1.record(ctx)
            ctx.synchronize()
            return e1.elapsed_ms_since(e0) / k
        one = group(20)
        k = max(10, min(400, int(150.0 / max(one, 1e-3))))
        group(k)
        k = max(10, min(200, int(40.0 / max(one, 1e-3))))
        return float(np.median([group(k) for _ in range(5)]))
    r = {"natural_ms": round(t(0), 4), "inverse_ms": round(t(0, 1), 4), "bitrev_ms": round(t(1), 4)}
    r["natural_frac_of_8TBps"] = round(16.0 * batch * n / (r["natural_ms"] * 1e-3) / 8e12, 4)
    out[f"2^{log_n}"] = r
    buf.free()
print(json.dumps(out), flush=True)
