#!/usr/bin/env python3
"""Forward 2^20 batch NTT (64 columns) timed with HIP events for several column-chunk sizes
(PLONKY2_NTT_CHUNK_COLS): does the intermediate of the two passes stay in the Infinity Cache?
natural order goes through the scratch workspace; bit-reversed runs in place."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import plonky2_gpu_amd as pg
from plonky2_gpu_amd import _lib

ctx = pg.Context(0)
log_n, batch = 20, 64
n = 1 << log_n
rng = np.random.default_rng(1)
host = rng.integers(0, 0xFFFFFFFF00000001, size=(batch, n), dtype=np.uint64)
buf = pg.DeviceBuffer.from_host(ctx, host)
for order in (0, 1):
    for chunk in (1, 2, 3, 4, 6, 8, 12, 16, 32, 64):
        os.environ["PLONKY2_NTT_CHUNK_COLS"] = str(chunk)
        ms = []
        for r in range(6):
            e0, e1 = pg.Event(), pg.Event()
            e0.record(ctx)
            _lib.call("gl_ntt_batch", buf.ptr, batch, log_n, n, 0, order, ctx.ptr)
            e1.record(ctx)
            ctx.synchronize()
            if r:
                ms.append(e1.elapsed_ms_since(e0))
        print(json.dumps({"order": "natural" if order == 0 else "bit-reversed", "chunk_cols": chunk, "fwd_ms": float(np.median(ms)),
                          "frac_of_8TBps": 16.0 * n * batch / (float(np.median(ms)) * 1e-3) / 8e12}), flush=True)
