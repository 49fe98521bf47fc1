"""Child process of test_gpu_ntt.py::test_alternative_kernel_selections: the environment selects another set of NTT
kernels (read once per process by the library), the transforms must equal the oracle's all the same."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle  # noqa: E402

import plonky2_gpu_amd as pg  # noqa: E402


def bitrev_perm(bits):
    idx = np.arange(1 << bits, dtype=np.uint64)
    out = np.zeros_like(idx)
    for b in range(bits):
        out |= ((idx >> np.uint64(b)) & np.uint64(1)) << np.uint64(bits - 1 - b)
    return out.astype(np.int64)


def main():
    gpu = pg.Context(0)
    for log_n in (9, 13, 16, 20, 21, 22):
        x = oracle.random_field((3, 1 << log_n), seed=9100 + log_n)
        exp = oracle.canon(oracle.fft_batch(x, threads=2))
        f = pg.fft_with_options(gpu, x)
        assert (f == exp).all(), ("forward", log_n)
        assert (pg.ifft_with_options(gpu, f) == x).all(), ("inverse", log_n)
        b = pg.fft_with_options(gpu, x[1], bit_reversed=True)
        assert (b == exp[1][bitrev_perm(log_n)]).all(), ("bit-reversed", log_n)
    for log_n, rate_bits in ((14, 3), (20, 1), (21, 1), (22, 0)):
        c = oracle.random_field((2, 1 << log_n), seed=9200 + log_n)
        got = pg.coset_lde_bit_reversed(gpu, c, rate_bits)
        perm = bitrev_perm(log_n + rate_bits)
        for i in range(2):
            assert (got[i] == oracle.canon(oracle.coset_lde(c[i], rate_bits))[perm]).all(), ("lde", log_n)
    print("ok")


if __name__ == "__main__":
    main()
