# TIMING ONLY: a 64-column batch as two 32-column halves on two streams (two contexts of one device) against one call on one stream.
# The halves share the library's one workspace, so the split run's RESULTS ARE WRONG; the question is only whether the kernels of two
# independent halves fill each other's start-up and drain (one workgroup per CU: they cannot share a CU).
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import plonky2_gpu_amd as pg
from plonky2_gpu_amd import _lib
a, b = pg.Context(0), pg.Context(0)
log_n, batch = 20, 64
n = 1 << log_n
host = np.random.default_rng(1).integers(0, 0xFFFFFFFF00000001, size=(batch, n), dtype=np.uint64)
buf = pg.DeviceBuffer.from_host(a, host)
half = batch // 2
def full(k):
    for _ in range(k):
        _lib.call("gl_ntt_batch", buf.ptr, batch, log_n, n, 0, 0, a.ptr)
def split(k):
    for _ in range(k):
        _lib.call("gl_ntt_batch", buf.ptr, half, log_n, n, 0, 0, a.ptr)
        _lib.call("gl_ntt_batch", buf.ptr + 8 * half * n, half, log_n, n, 0, 0, b.ptr)
def halves_one_stream(k):
    for _ in range(k):
        _lib.call("gl_ntt_batch", buf.ptr, half, log_n, n, 0, 0, a.ptr)
        _lib.call("gl_ntt_batch", buf.ptr + 8 * half * n, half, log_n, n, 0, 0, a.ptr)
def timed(fn, k):
    a.synchronize(); b.synchronize()
    t = time.perf_counter()
    fn(k)
    a.synchronize(); b.synchronize()
    return (time.perf_counter() - t) / k * 1e3
out = {}
for name, fn in (("one_call_one_stream", full), ("two_halves_two_streams", split), ("two_halves_one_stream", halves_one_stream)):
    timed(fn, 300)
    out[name] = [round(timed(fn, 400), 4) for _ in range(3)]
print(json.dumps({"ms_per_64_column_forward_batch": out, "note": "wall clock over 400 batches, steady state; the split runs share one workspace: timing only"}))
