import sys, os, ctypes, time
sys.path.insert(0, os.getcwd())
import numpy as np
import plonky2_gpu_amd._lib as L
variant = sys.argv[1]
if variant == "ablate":
    L.LIB_PATH = os.path.join(os.getcwd(), "plonky2_gpu_amd", "libplonky2_hip_ablate.so")
import plonky2_gpu_amd as pg
ctx = pg.Context(0)
n, batch, log_n = 1 << 20, 64, 20
buf = pg.DeviceBuffer(ctx, batch * n)
L.call("gl_memset_zero", buf.ptr, batch * n * 8, ctx.ptr)
for inv in (0, 1):
    for _ in range(3):
        L.call("gl_ntt_batch", buf.ptr, batch, log_n, n, inv, 0, ctx.ptr)
    ctx.synchronize()
    e0, e1 = pg.Event(), pg.Event()
    e0.record(ctx)
    for _ in range(10):
        L.call("gl_ntt_batch", buf.ptr, batch, log_n, n, inv, 0, ctx.ptr)
    e1.record(ctx)
    ctx.synchronize()
    ms = e1.elapsed_ms_since(e0) / 10
    print(variant, "inverse" if inv else "forward", "%.3f ms per 64-col batch, %.2f us/NTT" % (ms, ms * 1e3 / 64))
