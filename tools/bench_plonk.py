#!/usr/bin/env python3
"""Timing of the permutation-argument stage at the reference's ed25519 shape:
n = 2^18, 234 wires / 80 routed, 8 constants, 2 challenges, quotient_degree_factor 8, rate 8.
Random data (timing only; parity is covered by tests/test_gpu_plonk.py)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import plonky2_gpu_amd as pg
from plonky2_gpu_amd import _lib, gate_program as gp

P = pg.P


def rnd(rng, shape):
    a = rng.integers(0, 2**64, size=shape, dtype=np.uint64)
    return np.where(a >= np.uint64(P), a - np.uint64(P), a)


def timed(ctx, f, reps=3):
    f()
    ctx.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = pg.Event(), pg.Event()
        e0.record(ctx)
        f()
        e1.record(ctx)
        ctx.synchronize()
        best = min(best, e1.elapsed_ms_since(e0))
    return best


def main():
    log_n, wires, routed, consts, qdf, rate = 18, 234, 80, 8, 8, 3
    n = 1 << log_n
    ctx = pg.Context(0)
    rng = np.random.Generator(np.random.PCG64(1))
    d_w = pg.DeviceBuffer.from_host(ctx, rnd(rng, (wires, n)))
    d_cs = pg.DeviceBuffer.from_host(ctx, rnd(rng, (consts + routed, n)))
    d_k = pg.DeviceBuffer.from_host(ctx, np.array([pow(7, j, P) for j in range(routed)], dtype=np.uint64))
    betas, gammas, alphas = rnd(rng, 2), rnd(rng, 2), rnd(rng, 2)
    num_prods = -(-routed // qdf) - 1
    n_cols = 2 * (1 + num_prods)
    d_zpp = pg.DeviceBuffer(ctx, n_cols * n)
    b, g = np.ascontiguousarray(betas), np.ascontiguousarray(gammas)

    def pp():
        _lib.call("gl_permutation_partial_products", d_w.ptr, n, d_cs.at(consts * n), n, d_k.ptr, b.ctypes.data, g.ctypes.data, 2,
                  routed, qdf, log_n, d_zpp.ptr, ctx.ptr)

    t_pp = timed(ctx, pp)
    print(f"partial products + Z   n=2^{log_n} routed={routed} challenges=2: {t_pp:.3f} ms")
    wb = pg.PolynomialBatch.from_values_device(ctx, d_w, wires, log_n, rate, False, 4)
    cb = pg.PolynomialBatch.from_values_device(ctx, d_cs, consts + routed, log_n, rate, False, 4)
    zb = pg.PolynomialBatch.from_values_device(ctx, d_zpp, n_cols, log_n, rate, False, 4)
    out = [None]

    def quot(prog=None, ngc=0):
        def f():
            if out[0] is not None:
                out[0].free()
            out[0] = pg.compute_quotient_polys(ctx, wb, cb, zb, consts, routed, d_k, betas, gammas, alphas, qdf, None, ngc, prog)
        return f

    t_q = timed(ctx, quot())
    print(f"compute_quotient_polys (permutation terms only)  lde 2^{log_n + 3}: {t_q:.3f} ms")
    # a gate set of similar weight to a mid-size circuit: 8 arithmetic gates of 20 ops + constant + public input
    gates = [gp.noop_gate(), gp.constant_gate(2), gp.public_input_gate()] + [gp.arithmetic_gate(20)] * 3
    prog = pg.GateProgram(ctx, gates, [0] * 6, [(0, 6)], [1, 2, 3, 4])
    t_qg = timed(ctx, quot(prog, 20))
    print(f"compute_quotient_polys (+ 6 gate programs, {sum(len(x) for x in gates)} instructions): {t_qg:.3f} ms")
    pts = rnd(rng, (2, 2))
    t_ev = timed(ctx, lambda: wb.eval_polynomials_ext2(pts))
    print(f"openings: {wires} polys x 2^{log_n} at 2 ext points: {t_ev:.3f} ms (incl. D2H of the results)")


if __name__ == "__main__":
    main()
