# the coset LDE of configs[2] alone: 135 polynomials of 2^20 coefficients -> 8 cosets each (9 GB), no hashing around it
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import plonky2_gpu_amd as pg
from plonky2_gpu_amd import _lib
ctx = pg.Context(0)
log_n, cols = 20, int(os.environ.get("COLS", "135"))
n = 1 << log_n
seed = np.random.default_rng(1).integers(0, 0xFFFFFFFF00000001, size=16 * n, dtype=np.uint64)
src = pg.DeviceBuffer(ctx, cols * n)
for k in range(0, cols, 16):
    src.upload(seed[: min(16, cols - k) * n], k * n)
lde = pg.DeviceBuffer(ctx, cols * n * 8)
ms = []
for r in range(7):
    e0, e1 = pg.Event(), pg.Event()
    e0.record(ctx)
    _lib.call("gl_coset_lde_batch", src.ptr, lde.ptr, cols, log_n, 3, 7, n, n * 8, ctx.ptr)
    e1.record(ctx)
    ctx.synchronize()
    if r > 1: ms.append(e1.elapsed_ms_since(e0))
print(json.dumps({"coset_lde_%d_cols_2p20_rate8_ms" % cols: [round(float(np.median(ms)), 3), round(min(ms), 3)], "GBps_algorithmic": round(72.0 * n * cols / (np.median(ms) * 1e-3) / 1e9, 1)}), flush=True)
