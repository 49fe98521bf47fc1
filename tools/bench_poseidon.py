#!/usr/bin/env python3
"""Stand-alone Poseidon permutation rate: gl_poseidon_permute_batch on 2^22 states (the kernel the rocprofv3 counter
passes of profiles/ are taken on), HIP events. Prints one JSON line."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import plonky2_gpu_amd as pg
from plonky2_gpu_amd import _lib

count = 1 << 22
ctx = pg.Context(0)
rng = np.random.default_rng(3)
buf = pg.DeviceBuffer.from_host(ctx, rng.integers(0, 0xFFFFFFFF00000001, size=count * 12, dtype=np.uint64))
ms = []
for r in range(6):
    e0, e1 = pg.Event(), pg.Event()
    e0.record(ctx)
    _lib.call("gl_poseidon_permute_batch", buf.ptr, count, ctx.ptr)
    e1.record(ctx)
    ctx.synchronize()
    if r:
        ms.append(e1.elapsed_ms_since(e0))
m = float(np.median(ms))
print(json.dumps({"permutations": count, "ms": m, "permutations_per_s": count / (m * 1e-3)}))
