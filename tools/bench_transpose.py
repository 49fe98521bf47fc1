#!/usr/bin/env python3
"""Leaf-major copies: gl_transpose ([n_cols][col_stride] column-major -> [n_rows][n_cols]) timed with HIP events and checked against
numpy, for the leaf lengths of the paths that use it (135 = config #3; 234, 88, 20 = the ed25519 proof; 64, 7, 126, 127, 300 = edge
shapes of the strip kernel). PLONKY2_TRANSPOSE=tile selects the 64 x 64 tile kernel."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import plonky2_gpu_amd as pg
from plonky2_gpu_amd import _lib


def main():
    ctx = pg.Context(0)
    out = {"kernel": os.environ.get("PLONKY2_TRANSPOSE", "strip")}
    rng = np.random.default_rng(3)
    for n_cols, log_rows in ((135, 21), (234, 20), (88, 21), (20, 21), (64, 21), (7, 18), (126, 18), (127, 18), (300, 17), (135, 23)):
        n_rows = (1 << log_rows) - (5 if log_rows == 18 else 0)  # ragged last strip in the small cases
        stride = 1 << log_rows
        check = log_rows <= 21 and n_cols * stride <= (1 << 28)
        if check:
            host = rng.integers(0, pg.P, size=(n_cols, stride), dtype=np.uint64)
            d_c = pg.DeviceBuffer.from_host(ctx, host)
        else:
            d_c = pg.DeviceBuffer(ctx, n_cols * stride)
        d_r = pg.DeviceBuffer(ctx, n_cols * stride)
        ms = []
        for it in range(6):
            e0, e1 = pg.Event(), pg.Event()
            e0.record(ctx)
            _lib.call("gl_transpose", d_c.ptr, d_r.ptr, n_cols, n_rows, stride, ctx.ptr)
            e1.record(ctx)
            ctx.synchronize()
            if it:
                ms.append(e1.elapsed_ms_since(e0))
        ok = None
        if check:
            got = d_r.download(0, n_rows * n_cols).reshape(n_rows, n_cols)
            ok = bool((got == host[:, :n_rows].T).all())
            if not ok:
                print(json.dumps({"mismatch": [n_cols, log_rows]}))
                sys.exit(1)
        t = float(np.median(ms))
        out[f"{n_cols}x2^{log_rows}"] = {"ms": round(t, 3), "GBps": round(2 * 8 * n_cols * n_rows / t / 1e6, 1), "equal_to_numpy": ok}
        d_c.free()
        d_r.free()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
