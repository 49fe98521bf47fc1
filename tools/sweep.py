#!/usr/bin/env python3
"""NTTs/s and Merkle-leaves-hashed/s over trace sizes 2^16..2^23 on one GPU (north-star sweep).
NTT: batch of columns sized to 512 MiB, forward + inverse, natural order. Commit: from_values with
rate 8, cap height 4, leaf-major copy included: the three commitments of the ed25519 shape at 2^18 rows, and — since round 5 — the
FULL WIDTH of standard_recursion_config, 135 columns, at 2^20, 2^21, 2^22 and 2^23 rows (LDE 9 / 18 / 36 / 72 GB; the last two hold
more than 2^32 elements), each with its fraction of the HBM roofline (algorithmic bytes of SURVEY 8d / time / 8 TB/s). HIP-event timing."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import plonky2_gpu_amd as pg
from plonky2_gpu_amd import _lib


def fill(ctx, buf, n_elems, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    chunk = 1 << 24
    for off in range(0, n_elems, chunk):
        k = min(chunk, n_elems - off)
        a = rng.integers(0, 2**64, size=k, dtype=np.uint64)
        buf.upload(np.where(a >= np.uint64(pg.P), a - np.uint64(pg.P), a), off)


def main():
    ctx = pg.Context(0)
    rows = []
    for log_n in range(16, 24):
        n = 1 << log_n
        batch = max(2, (1 << 26) >> log_n)  # 512 MiB
        buf = pg.DeviceBuffer(ctx, batch * n)
        fill(ctx, buf, batch * n, log_n)
        def step():
            _lib.call("gl_ntt_batch", buf.ptr, batch, log_n, n, 0, 0, ctx.ptr)
            _lib.call("gl_ntt_batch", buf.ptr, batch, log_n, n, 1, 0, ctx.ptr)
        step(); ctx.synchronize()
        e0, e1 = pg.Event(), pg.Event()
        reps = 5
        e0.record(ctx)
        for _ in range(reps):
            step()
        e1.record(ctx); ctx.synchronize()
        ms = e1.elapsed_ms_since(e0) / reps
        r = {"kind": "ntt", "log_n": log_n, "batch": batch, "ntts_per_s": 2 * batch / (ms * 1e-3),
             "alg_GBps": 2 * batch * 16.0 * n / (ms * 1e-3) / 1e9}
        rows.append(r); print(json.dumps(r)); buf.free()
    for log_n, cols in [(18, 234), (18, 20), (18, 16), (20, 135), (21, 135), (22, 135), (23, 135)]:
        n, n_ext = 1 << log_n, 1 << (log_n + 3)
        d_vals = pg.DeviceBuffer(ctx, cols * n); fill(ctx, d_vals, cols * n, 100 + log_n)
        d_work = pg.DeviceBuffer(ctx, cols * n)
        d_lde = pg.DeviceBuffer(ctx, cols * n_ext); d_leaves = pg.DeviceBuffer(ctx, cols * n_ext)
        d_dig = pg.DeviceBuffer(ctx, 8 * (n_ext - 16)); d_cap = pg.DeviceBuffer(ctx, 64)
        best = 1e9
        for it in range(3):
            _lib.call("gl_memcpy_d2d", d_work.ptr, d_vals.ptr, cols * n * 8, ctx.ptr); ctx.synchronize()
            e0, e1 = pg.Event(), pg.Event()
            e0.record(ctx)
            _lib.call("gl_commit_from_values", d_work.ptr, cols, log_n, 3, 4, 0, 7, d_lde.ptr, d_leaves.ptr, d_dig.ptr, d_cap.ptr, ctx.ptr)
            e1.record(ctx); ctx.synchronize()
            if it: best = min(best, e1.elapsed_ms_since(e0))
        perms = n_ext * ((cols + 7) // 8) + n_ext - 16
        alg = 8.0 * cols * n + 8.0 * cols * n_ext + 32.0 * (2 * (n_ext - 16) + 16)
        r = {"kind": "commit", "log_n": log_n, "cols": cols, "lde_elements": cols * n_ext, "lde_GB": cols * n_ext * 8 / 1e9, "ms": best,
             "leaves_per_s": n_ext / (best * 1e-3), "permutations_per_s": perms / (best * 1e-3),
             "algorithmic_GBps": alg / (best * 1e-3) / 1e9, "hbm_frac": alg / (best * 1e-3) / 1e9 / 8000.0}
        rows.append(r); print(json.dumps(r))
        for b in (d_vals, d_work, d_lde, d_leaves, d_dig, d_cap): b.free()


if __name__ == "__main__":
    main()
