"""Pins the C oracle's FFT family with the reference's own test properties
(field/src/fft.rs:242-309, field/src/polynomial/mod.rs:482-522, plonky2/src/util/mod.rs:70-166)."""
import numpy as np

from oracle import pyref

P = pyref.P


def test_reverse_bits_known_answers(oracle):
    L = oracle.lib()
    # plonky2/src/util/mod.rs:70-76
    assert L.glo_reverse_bits(0b0000000000, 10) == 0b0000000000
    assert L.glo_reverse_bits(0b0000000001, 10) == 0b1000000000
    assert L.glo_reverse_bits(0b1000000000, 10) == 0b0000000001
    assert L.glo_reverse_bits(0b00000, 5) == 0b00000
    assert L.glo_reverse_bits(0b01011, 5) == 0b11010
    # util/mod.rs:78-102 and util/src/lib.rs tests: in-place permutation
    v = np.array([10, 20, 30, 40], dtype=np.uint64)
    L.glo_reverse_index_bits_in_place(v.ctypes.data_as(oracle._u64p), 4)
    assert v.tolist() == [10, 30, 20, 40]
    v = np.arange(256, dtype=np.uint64)
    L.glo_reverse_index_bits_in_place(v.ctypes.data_as(oracle._u64p), 256)
    assert v.tolist() == [int(f"{i:08b}"[::-1], 2) for i in range(256)]
    # the first 16 entries of the reference's literal table (util/mod.rs:82-85)
    assert v[:16].tolist() == [0, 128, 64, 192, 32, 160, 96, 224, 16, 144, 80, 208, 48, 176, 112, 240]


def test_fft_equals_naive_evaluation(oracle):
    # fft.rs:252-282: degree 200, coeffs i*1337 % 100, padded to 256
    degree, n = 200, 256
    coeffs = [(i * 1337) % 100 for i in range(degree)] + [0] * (n - degree)
    points = oracle.canon(oracle.fft(coeffs))
    assert points.tolist() == pyref.dft(coeffs)
    back = oracle.canon(oracle.ifft(points))
    assert back.tolist() == coeffs
    # zero_factor r = 0..3 equals the plain fft of the zero-padded polynomial
    for r in range(4):
        tail = coeffs + [0] * (n * ((1 << r) - 1))
        a = oracle.canon(oracle.fft(tail))
        b = oracle.canon(oracle.fft(tail, r=r))
        assert (a == b).all()


def test_fft_random_sizes(oracle):
    for lg in range(1, 9):
        x = oracle.random_field(1 << lg, seed=lg)
        assert oracle.canon(oracle.fft(x)).tolist() == pyref.dft(x.tolist())
        assert oracle.canon(oracle.ifft(x)).tolist() == pyref.idft(x.tolist())
    x = oracle.random_field(1 << 12, seed=99)
    assert oracle.canon(oracle.fft(x)).tolist() == pyref.fast_ntt(x.tolist())
    assert oracle.canon(oracle.ifft(x)).tolist() == pyref.fast_ntt(x.tolist(), inverse=True)


def test_pyref_fast_ntt_matches_definition():
    g = pyref.splitmix64(5)
    x = [next(g) for _ in range(64)]
    assert pyref.fast_ntt(x) == pyref.dft(x)
    assert pyref.fast_ntt(x, inverse=True) == pyref.idft(x)


def test_coset_fft_and_ifft(oracle):
    # polynomial/mod.rs:482-522: random degree-256 poly, shift = generator, naive coset evaluation
    n = 256
    x = oracle.random_field(n, seed=7)
    shift = 7
    ev = oracle.canon(oracle.coset_fft(x, shift)).tolist()
    w = pyref.root_of_unity(8)
    naive = [sum(int(c) * pow(shift * pow(w, k, P) % P, j, P) for j, c in enumerate(x)) % P for k in range(n)]
    assert ev == naive
    assert oracle.canon(oracle.coset_ifft(np.array(ev, dtype=np.uint64), shift)).tolist() == x.tolist()
    assert pyref.coset_idft(ev, shift) == x.tolist()


def test_coset_lde(oracle):
    for lg, rate in [(3, 3), (5, 3), (4, 1), (6, 2), (4, 0)]:
        c = oracle.random_field(1 << lg, seed=lg * 10 + rate)
        got = oracle.canon(oracle.coset_lde(c, rate)).tolist()
        assert got == pyref.coset_lde(c.tolist(), rate)


def test_root_table_layout(oracle):
    # fft.rs:15-34: row s = powers of w_{2^(s+1)}, max(2^s, 2) entries; concat has n entries for n>=4
    n = 64
    t = oracle.canon(oracle.root_table_concat(n)).tolist()
    exp = []
    for s in range(6):
        w = pyref.root_of_unity(s + 1)
        exp += [pow(w, i, P) for i in range(max(1 << s, 2))]
    assert t == exp and len(t) == n
