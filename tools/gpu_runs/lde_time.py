import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import plonky2_gpu_amd as pg
from plonky2_gpu_amd import _lib
ctx = pg.Context(0)
out = {"tag": os.environ.get("TAG"), "direct": os.environ.get("PLONKY2_NTT_DIRECT"), "note": "ms per call, [median, min] of 12"}
def timed(fn):
    ms = []
    for r in range(14):
        e0, e1 = pg.Event(), pg.Event()
        e0.record(ctx)
        fn()
        e1.record(ctx)
        ctx.synchronize()
        if r > 1: ms.append(e1.elapsed_ms_since(e0))
    return [round(float(np.median(ms)), 4), round(float(min(ms)), 4)]
for log_n in [int(x) for x in os.environ.get("SIZES", "18,20,21").split(",")]:
    n = 1 << log_n
    batch = (1 << 26) >> log_n
    rng = np.random.default_rng(1)
    host = rng.integers(0, 0xFFFFFFFF00000001, size=(batch, n), dtype=np.uint64)
    buf = pg.DeviceBuffer.from_host(ctx, host)
    r = {"bitrev_512MiB_ms": timed(lambda: _lib.call("gl_ntt_batch", buf.ptr, batch, log_n, n, 0, 1, ctx.ptr))}
    cols = batch // 8
    lde = pg.DeviceBuffer(ctx, cols * n * 8)
    r["coset_lde_rate8_%dcols_ms" % cols] = timed(lambda: _lib.call("gl_coset_lde_batch", buf.ptr, lde.ptr, cols, log_n, 3, 7, n, n * 8, ctx.ptr))
    r["coset_lde_algorithmic_frac_of_8TBps"] = round((8.0 * cols * n * 9) / (r["coset_lde_rate8_%dcols_ms" % cols][0] * 1e-3) / 8e12, 4)
    out[f"2^{log_n}"] = r
    buf.free(); lde.free()
print(json.dumps(out), flush=True)
