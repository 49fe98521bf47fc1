"""Differential campaign one level below fuzz_prove.py: randomly shaped NTTs, coset LDEs and
PolynomialBatch commits on the device against the C oracle (oracle/gl_oracle.c).
    python tests/fuzz_commit.py [cases=60] [seed=1] [large]
`large`: 2^13..2^22 rows (every two-pass decomposition, wide and narrow tiles, ragged tiles, the split columns of 2^21, three passes at 2^22),
1..5 polynomials, rate 0..1.
Shapes: 1..48 polynomials (crossing the 8-element sponge block and the <=4 `not hashed` rule), 2^0..2^13
rows, rate 0..3 bits, every legal cap height, values that are NOT canonical (>= p) in a tenth of the
cases, from_values and from_coeffs, with and without the leaf-major copy; forward / inverse NTT with
natural and bit-reversed output. Exits non-zero on the first mismatch."""
import json
import os
import random
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import plonky2_gpu_amd as pg  # noqa: E402
from oracle import oracle as o  # noqa: E402
from plonky2_gpu_amd import _lib  # noqa: E402


def rand_values(nprng, shape, non_canonical):
    if non_canonical:  # any u64 is a legal representative at the boundary (goldilocks_field.rs:26)
        return nprng.integers(0, 2**64, size=shape, dtype=np.uint64)
    return nprng.integers(0, pg.P, size=shape, dtype=np.uint64)


def bitrev_perm(n):
    bits = n.bit_length() - 1
    idx = np.arange(n, dtype=np.int64)
    out = np.zeros(n, dtype=np.int64)
    for b in range(bits):
        out |= ((idx >> b) & 1) << (bits - 1 - b)
    return out


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    large = len(sys.argv) > 3 and sys.argv[3] == "large"
    rng, nprng = random.Random(seed), np.random.default_rng(seed)
    ctx = pg.Context(0)
    t0 = time.time()
    for k in range(cases):
        if large:
            log_n = rng.choice([13, 14, 15, 16, 17, 18, 19, 20, 21, 22])
            n_polys = rng.choice([1, 2, 3, 5])
            rate_bits = rng.choice([0, 1]) if log_n < 21 else 0  # the LDE of 2^21 and 2^22 rows is in test_gpu_ntt.py
            cap_height = rng.randrange(0, 6)
        else:
            log_n = rng.choice([0, 1, 2, 3, 5, 7, 9, 10, 11, 12, 13])
            n_polys = rng.choice([1, 2, 3, 4, 5, 7, 8, 9, 12, 16, 17, 31, 48])
            rate_bits = rng.choice([0, 1, 2, 3])
            cap_height = rng.randrange(0, log_n + rate_bits + 1)
        non_canonical = rng.random() < 0.1
        from_values = rng.random() < 0.5
        leaf_major = rng.random() < 0.5
        vals = rand_values(nprng, (n_polys, 1 << log_n), non_canonical)
        desc = dict(log_n=log_n, n_polys=n_polys, rate_bits=rate_bits, cap_height=cap_height, non_canonical=non_canonical,
                    from_values=from_values, leaf_major=leaf_major)
        # ---- commit
        if from_values:
            b = pg.PolynomialBatch.from_values(ctx, vals, rate_bits, False, cap_height, leaf_major=leaf_major)
            exp = o.commit_from_values(vals, rate_bits, cap_height, threads=4)
        else:
            b = pg.PolynomialBatch.from_coeffs(ctx, vals, rate_bits, False, cap_height, leaf_major=leaf_major)
            exp = o.commit_from_coeffs(vals, rate_bits, cap_height, threads=4)
        ok = (b.merkle_tree.cap == o.canon(exp["cap"])).all() and (b.merkle_tree.digests == o.canon(exp["digests"]).reshape(-1, 4)).all()
        if from_values:
            ok = ok and (b.polynomials == o.canon(exp["coeffs"])).all()
        leaves = o.canon(exp["leaves"])
        ok = ok and (b.lde_column_major() == leaves.T).all()
        if leaf_major:
            ok = ok and (b.merkle_tree.d_leaves.download().reshape(leaves.shape) == leaves).all()
        idx = [rng.randrange(leaves.shape[0]) for _ in range(3)]
        lv, sib = b.merkle_tree.open_batch(idx)
        for q, i in enumerate(idx):
            ok = ok and (lv[q] == leaves[i]).all() and o.merkle_verify(leaves[i], i, exp["cap"], sib[q])
        # ---- NTT on the same data: forward natural, forward bit-reversed, inverse round trip
        x = vals.copy()
        nat = pg.fft_with_options(ctx, x)
        want = o.canon(o.fft_batch(x.copy(), threads=4))
        ok = ok and (nat == want).all()
        buf = pg.DeviceBuffer.from_host(ctx, x)
        n = 1 << log_n
        _lib.call("gl_ntt_batch", buf.ptr, n_polys, log_n, n, 0, 1, ctx.ptr)  # bit-reversed output
        ok = ok and (buf.download().reshape(n_polys, n) == want[:, bitrev_perm(n)]).all()
        buf.upload(want)
        _lib.call("gl_ntt_batch", buf.ptr, n_polys, log_n, n, 1, 0, ctx.ptr)  # inverse, natural
        ok = ok and (buf.download().reshape(n_polys, n) == o.canon(x)).all()
        print(f"case {k:3d} {'ok  ' if ok else 'FAIL'} {desc}", flush=True)
        if not ok:
            print(json.dumps(dict(failed_case=k, seed=seed, **desc)))
            sys.exit(1)
    print(json.dumps(dict(cases=cases, seed=seed, all_equal=True, seconds=round(time.time() - t0, 1))))


if __name__ == "__main__":
    main()
