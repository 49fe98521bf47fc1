# bit-reversed and natural 2^20 transforms at batch sizes from 0.5 to 8 GiB: ms per 64 polynomials (does the rate hold when the batch
# no longer fits the 256 MiB memory-side cache?)
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import plonky2_gpu_amd as pg
from plonky2_gpu_amd import _lib
ctx = pg.Context(0)
log_n = 20
n = 1 << log_n
out = {"note": "2^20-point transforms, ms per 64 polynomials [median of 6]", "data": os.environ.get("DATA", "random")}
for batch in [int(x) for x in os.environ.get("BATCHES", "64,256,1024").split(",")]:
    buf = pg.DeviceBuffer(ctx, batch * n)
    if os.environ.get("DATA", "random") == "random":  # 64 random polynomials, copied over the whole batch (zeros run at a higher clock)
        seed = pg.DeviceBuffer.from_host(ctx, np.random.default_rng(1).integers(0, 0xFFFFFFFF00000001, size=64 * n, dtype=np.uint64))
        for k in range(0, batch, 64):
            _lib.call("gl_memcpy_d2d", buf.ptr + k * n * 8, seed.ptr, 64 * n * 8, ctx.ptr)
        ctx.synchronize()
        seed.free()
    def t(order, inverse=0):
        ms = []
        for r in range(8):
            e0, e1 = pg.Event(), pg.Event()
            e0.record(ctx)
            _lib.call("gl_ntt_batch", buf.ptr, batch, log_n, n, inverse, order, ctx.ptr)
            e1.record(ctx)
            ctx.synchronize()
            if r > 1: ms.append(e1.elapsed_ms_since(e0))
        return round(float(np.median(ms)) * 64 / batch, 4)
    out[str(batch)] = {"bit_reversed_in_place": t(1), "natural": t(0)}
    buf.free()
print(json.dumps(out), flush=True)
