"""Probe (round 6): what a SECOND context's small operations wait for while a first context runs a long commit. Prints, per step of the
second context (upload, 2^12 transform, download), how long it took alone and during the commit, and when it finished relative to the commit."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np

import plonky2_gpu_amd as pg
from plonky2_gpu_amd import _lib

a, b = pg.Context(0), pg.Context(0)
cols, log_n = 135, 19
n, n_ext = 1 << log_n, 1 << (log_n + 3)
rng = np.random.default_rng(1)
d_vals = pg.DeviceBuffer.from_host(a, rng.integers(0, 2**63, size=cols * n, dtype=np.uint64))
d_work, d_lde = pg.DeviceBuffer(a, cols * n), pg.DeviceBuffer(a, cols * n_ext)
d_dig, d_cap = pg.DeviceBuffer(a, 8 * (n_ext - 16)), pg.DeviceBuffer(a, 64)
small = rng.integers(0, 2**63, size=2 << 12, dtype=np.uint64)
d_small = pg.DeviceBuffer(b, small.size)
pinned = pg.PinnedArray(small.size)
pinned.array[:] = small


def commit():
    _lib.call("gl_memcpy_d2d", d_work.ptr, d_vals.ptr, cols * n * 8, a.ptr)
    _lib.call("gl_commit_from_values", d_work.ptr, cols, log_n, 3, 4, 0, 7, d_lde.ptr, None, d_dig.ptr, d_cap.ptr, a.ptr)


def steps(kind):
    out = {}
    t = time.perf_counter()
    if kind == "pinned":
        _lib.call("gl_memcpy_h2d", d_small.ptr, pinned.ptr, small.size * 8, b.ptr)
    else:
        d_small.upload(small)
    out["h2d_ms"] = (time.perf_counter() - t) * 1e3
    t = time.perf_counter()
    _lib.call("gl_ntt_batch", d_small.ptr, 2, 12, 1 << 12, 0, 0, b.ptr)
    b.synchronize()
    out["ntt_ms"] = (time.perf_counter() - t) * 1e3
    t = time.perf_counter()
    d_small.download()
    out["d2h_ms"] = (time.perf_counter() - t) * 1e3
    return {k: round(v, 3) for k, v in out.items()}


commit(); a.synchronize(); steps("pageable"); steps("pinned")
res = {"GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES"), "alone": steps("pageable"), "alone_pinned": steps("pinned")}
for kind in ("pageable", "pinned"):
    a.synchronize()
    t0 = time.perf_counter()
    commit()
    tq = time.perf_counter()
    s = steps(kind)
    ts = time.perf_counter()
    a.synchronize()
    tc = time.perf_counter()
    res["during_commit_" + kind] = dict(s, queued_ms=round((tq - t0) * 1e3, 3), small_done_ms=round((ts - t0) * 1e3, 3), commit_done_ms=round((tc - t0) * 1e3, 3))
print(json.dumps(res))
