#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the independent Python big-int model (oracle/pyref.py).

The reference holds no golden FFT outputs, Merkle caps or commitments, so the
fixtures are produced here, in the build container, by the model that is itself pinned by the
reference's known answers (tests/test_oracle_*.py). Inputs come from the SplitMix64 stream of
SURVEY.md §8d, seed 0x706C6F6E6B7932. Run:  python tests/golden/gen_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyref  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
SEED = 0x706C6F6E6B7932


def u64(a):
    return np.array(a, dtype=np.uint64)


def main():
    os.makedirs(OUT, exist_ok=True)
    g = pyref.splitmix64(SEED)
    # NTT / inverse NTT / coset LDE (rate 8, shift 7), natural order
    for lg in (4, 8, 12):
        n = 1 << lg
        x = [next(g) for _ in range(n)]
        fwd = pyref.fast_ntt(x)
        inv = pyref.fast_ntt(x, inverse=True)
        lde_n = min(n, 1 << 9)
        scaled = [c * pow(7, i, pyref.P) % pyref.P for i, c in enumerate(x[:lde_n])] + [0] * (7 * lde_n)
        lde = pyref.fast_ntt(scaled)
        np.savez_compressed(os.path.join(OUT, f"ntt_2e{lg}.npz"), x=u64(x), fft=u64(fwd), ifft=u64(inv),
                            lde_coeffs=u64(x[:lde_n]), lde_rate8_natural=u64(lde))
        print("ntt", lg)
    # Merkle trees: leaf lengths <= 4 (no-op hash), not a multiple of 8, exactly 8, > 8; several caps
    for (n, k, h) in [(16, 3, 2), (16, 4, 0), (32, 7, 1), (16, 8, 4), (32, 20, 3), (8, 135, 1)]:
        leaves = [[next(g) for _ in range(k)] for _ in range(n)]
        dig, cap = pyref.merkle_tree(leaves, h)
        np.savez_compressed(os.path.join(OUT, f"merkle_n{n}_k{k}_h{h}.npz"), leaves=u64(leaves),
                            digests=u64(dig).reshape(-1, 4), cap=u64(cap).reshape(-1, 4), cap_height=h)
        print("merkle", n, k, h)
    # one full commit: from_values, n = 2^6, 135 columns, rate 8, cap height 4 (BASELINE configs[2] shape, small)
    n_polys, lg, rate, h = 135, 6, 3, 4
    vals = [[next(g) for _ in range(1 << lg)] for _ in range(n_polys)]
    coeffs, leaves, dig, cap = pyref.commit_from_values(vals, rate, h)
    np.savez_compressed(os.path.join(OUT, "commit_p135_2e6_r3_h4.npz"), values=u64(vals), coeffs=u64(coeffs),
                        leaves=u64(leaves), digests=u64(dig).reshape(-1, 4), cap=u64(cap).reshape(-1, 4),
                        rate_bits=rate, cap_height=h)
    print("commit")


if __name__ == "__main__":
    main()
