"""gl_field.h against Python big ints on the reference's edge operands
(field/src/prime_field_testing.rs:7-17, 79-125). Bit-exact after canonicalisation."""
import itertools

import numpy as np
import pytest

from gpu_util import P, gpu  # noqa: F401

pytestmark = pytest.mark.gpu


def edge_operands():
    base = list(range(0, 10))
    for c in (1 << 31, 1 << 32, 1 << 63):
        base += list(range(c - 10, c + 11))
    base += list(range(P - 10, P))
    base += [P, P + 1, (1 << 64) - 1, (1 << 64) - 2, P + (1 << 31)]
    return sorted(set(base))


def run_op(gpu, op, a, b=None):
    import plonky2_gpu_amd as pg
    from plonky2_gpu_amd import _lib

    da = pg.DeviceBuffer.from_host(gpu, a)
    db = pg.DeviceBuffer.from_host(gpu, b) if b is not None else None
    do = pg.DeviceBuffer(gpu, len(a))
    _lib.call("gl_debug_field_op", op, da.ptr, db.ptr if db else None, do.ptr, len(a), gpu.ptr)
    return do.download().tolist()


def test_binary_ops(gpu):
    ops = edge_operands()
    pairs = list(itertools.product(ops, ops))
    a = np.array([x for x, _ in pairs], dtype=np.uint64)
    b = np.array([y for _, y in pairs], dtype=np.uint64)
    assert run_op(gpu, 0, a, b) == [(x + y) % P for x, y in pairs]
    assert run_op(gpu, 1, a, b) == [(x - y) % P for x, y in pairs]
    assert run_op(gpu, 2, a, b) == [(x * y) % P for x, y in pairs]
    assert run_op(gpu, 5, a, b) == [(x + y * y) % P for x, y in pairs]
    assert run_op(gpu, 7, a, b) == [(x + y) % P for x, y in pairs]


def test_unary_ops_and_shifts(gpu):
    ops = edge_operands()
    rng = np.random.default_rng(1)
    ops += rng.integers(0, 2**64, size=500, dtype=np.uint64).tolist()
    a = np.array(ops, dtype=np.uint64)
    assert run_op(gpu, 3, a) == [(-x) % P for x in ops]
    assert run_op(gpu, 4, a) == [pow(x, 7, P) for x in ops]
    for k in range(192):  # x * 2^k for every shift the radix butterflies can use
        kk = np.full(len(ops), k, dtype=np.uint64)
        assert run_op(gpu, 6, a, kk) == [(x << k) % P for x in ops], k


def test_canonical_domain_asm_primitives(gpu):
    """add_c / sub_c / canon_c / mul_c / reduce128_c (hand-written carry chains) on the edge
    operands and on random data; outputs must already be canonical (no fix-up on the way out)."""
    ops = edge_operands()
    pairs = list(itertools.product(ops, ops))
    rng = np.random.default_rng(7)
    rnd = rng.integers(0, 2**64, size=(20000, 2), dtype=np.uint64).tolist()
    pairs += [(int(x), int(y)) for x, y in rnd]
    a = np.array([x for x, _ in pairs], dtype=np.uint64)
    b = np.array([y for _, y in pairs], dtype=np.uint64)
    assert run_op(gpu, 8, a, b) == [(x + y) % P for x, y in pairs]
    assert run_op(gpu, 9, a, b) == [(x - y) % P for x, y in pairs]
    assert run_op(gpu, 10, a, b) == [(x * y) % P for x, y in pairs]
    assert run_op(gpu, 11, a) == [x % P for x, _ in pairs]
    M = (1 << 64) - 1
    exp = []
    for x, y in pairs:
        prod = x * y
        lo, hi = (prod & M) ^ y, (prod >> 64) ^ x
        exp.append(((hi << 64) | lo) % P)
    assert run_op(gpu, 12, a, b) == exp


def _dotacc(mode, x, y):
    """the accumulator capi.hip's dotacc_from() builds from two test words"""
    m64 = (1 << 64) - 1
    if mode == 0:
        return x, y, ((~x & m64) + ((y << 13) & m64)) & m64, y >> 59, x >> 58, (x ^ y) & 7
    return x, y & 0xFFFF, (y >> 16) & 0xFFFF, (y >> 32) & 0xFF, (y >> 40) & 0xFF, (y >> 48) & 0xFF


def _dot_value(a0, a1, a2, k0, k1, k2):
    return (a0 + (a1 << 32) + (a2 << 64) + (k0 << 64) + (k1 << 96) + (k2 << 128)) % P


def test_dot_finish_including_its_rare_wrap_corrections(gpu):
    """gl::dot_finish (carry flags) and dot_finish_generic against big ints. Two of the three wrap
    corrections fire with probability ~2^-32 on random accumulators, so they are forced here:
      borrow of (w0,w1) - w3:  low words zero, a2 = 2^64-1 (mode 0: x = 0, y = 0)
      borrow of r.hi - w4:     everything zero except k2 (mode 1: x = 0, y = k2 << 48)"""
    ops = edge_operands()
    rng = np.random.default_rng(3)
    pairs = list(itertools.product(ops, ops))
    pairs += [(int(u), int(v)) for u, v in rng.integers(0, 2**64, size=(20000, 2), dtype=np.uint64)]
    forced = [(0, 0), (0, 5 << 48), (0, 255 << 48), (3, 1 << 48), (0, (200 << 40) | (7 << 48)), ((1 << 64) - 1, (255 << 32) | (255 << 48)),
              (0, 0xFFFF | (0xFFFF << 16) | (255 << 32) | (255 << 40) | (255 << 48))]
    pairs += forced
    a = np.array([x for x, _ in pairs], dtype=np.uint64)
    b = np.array([y for _, y in pairs], dtype=np.uint64)
    for mode, op_asm, op_generic in ((0, 13, 14), (1, 15, 16)):
        exp = [_dot_value(*_dotacc(mode, x, y)) for x, y in pairs]
        assert run_op(gpu, op_generic, a, b) == exp
        assert run_op(gpu, op_asm, a, b) == exp
    # the crafted inputs do reach the rare paths (so the test keeps covering them)
    a0, a1, a2, k0, k1, k2 = _dotacc(0, 0, 0)
    w3 = (a2 >> 32) + k1
    assert (a0 & 0xFFFFFFFF) + (((a0 >> 32) + (a1 & 0xFFFFFFFF)) << 32) < w3          # first correction
    a0, a1, a2, k0, k1, k2 = _dotacc(1, 0, 5 << 48)
    assert a0 == a1 == a2 == k0 == k1 == 0 and k2 == 5                                   # last correction: 0 - 5*2^32


def test_fold96_over_its_whole_stated_domain(gpu):
    """gl::fold96(al, ah) = al + ah * 2^32 mod p for ANY al and ah < 2^63 — the reduction of the gate programs' ACC
    accumulators (and of the Poseidon kernel's MDS column sums, where ah stays below 2^43). The edge operands put the
    inner carry, the single wrap of l + h * (2^32 - 1) and the follow-up correction at their extremes."""
    import random

    rng = random.Random(17)
    lows = edge_operands()
    highs = sorted(set([0, 1, 2, (1 << 31) - 1, 1 << 31, (1 << 32) - 1, 1 << 32, (1 << 32) + 1, (1 << 43) - 1, 1 << 43, (1 << 62) - 1,
                        1 << 62, (1 << 63) - (1 << 32), (1 << 63) - (1 << 32) + 1, (1 << 63) - 2, (1 << 63) - 1]
                       + [rng.randrange(1 << 63) for _ in range(40)]))
    pairs = list(itertools.product(lows, highs)) + [(rng.randrange(1 << 64), rng.randrange(1 << 63)) for _ in range(20000)]
    a = np.array([p[0] for p in pairs], dtype=np.uint64)
    b = np.array([p[1] for p in pairs], dtype=np.uint64)
    got = run_op(gpu, 17, a, b)
    for (x, y), g in zip(pairs, got):
        assert g == (x + (y << 32)) % P, (hex(x), hex(y))


def test_deferred_rare_paths(gpu):
    """add_f / sub_f / mul_f / mul_pow2_f + their *_fix (gl_field.h): the fast paths of a GROUP of operations, one branch, the
    corrections behind it — the form the NTT passes use. Every member of a group of three over the edge operands (both >= p: add's
    second wrap; b > p and a < 2^32: sub's second borrow; lo < hh + c1: mul's borrow, e.g. 2^63 * 2^63) and over random data."""
    ops = edge_operands()
    pairs = list(itertools.product(ops, ops))
    rng = np.random.default_rng(11)
    pairs += [(int(x), int(y)) for x, y in rng.integers(0, 2**64, size=(20000, 2), dtype=np.uint64).tolist()]
    # products whose low 64 bits are tiny (mul's borrow): x * y with y = the inverse-like partner making lo small
    pairs += [(1 << 63, 1 << 63), ((1 << 64) - 1, (1 << 64) - 1), ((1 << 32) + 1, (1 << 64) - (1 << 32)), (P - 1, P - 1), (1 << 32, 1 << 32)]
    a = np.array([x for x, _ in pairs], dtype=np.uint64)
    b = np.array([y for _, y in pairs], dtype=np.uint64)
    members = [lambda x, y: (x, y), lambda x, y: (y, x), lambda x, y: (x ^ y, x)]
    for which, m in enumerate(members):
        assert run_op(gpu, 18 + which, a, b) == [sum(m(x, y)) % P for x, y in pairs], ("add", which)
        assert run_op(gpu, 21 + which, a, b) == [(m(x, y)[0] - m(x, y)[1]) % P for x, y in pairs], ("sub", which)
        assert run_op(gpu, 24 + which, a, b) == [(m(x, y)[0] * m(x, y)[1]) % P for x, y in pairs], ("mul", which)
    xs = ops + rng.integers(0, 2**64, size=2000, dtype=np.uint64).tolist() + [0, (1 << 64) - 1, 1, (1 << 32) - 1, 1 << 32]
    ax = np.array(xs, dtype=np.uint64)
    M = (1 << 64) - 1
    for k in range(96):  # x 2^k + (~x) 2^k = (2^64 - 1) 2^k
        kk = np.full(len(xs), k, dtype=np.uint64)
        assert run_op(gpu, 27, ax, kk) == [(M << k) % P] * len(xs), k
        assert run_op(gpu, 27, ax, kk + np.uint64(1 << 32)) == [(x << k) % P for x in xs], k


def _dif_stage(v, s, d=4):
    """One radix-2 stage of the in-register DIF butterflies (ntt_kernels.h radix_dif_stage): blocks of 2^d slots, stage s."""
    half = 1 << s
    out = list(v)
    for base in range(0, 16, 1 << d):
        for bb in range((1 << d) // 2):
            i0 = base + (bb // half) * 2 * half + (bb % half)
            i1 = i0 + half
            k = (39 * (bb % half) * (32 >> s)) % 192
            a, c = v[i0], v[i1]
            out[i0] = (a + c) % P
            out[i1] = (a - c) * pow(2, k, P) % P
    return out


def test_radix_routines_with_operands_that_flag_their_deferred_corrections(gpu):
    """radix_dif_stage / radix_dif / radix_dif_blocks / shift_twiddles_radix4 (ntt_kernels.h) one vector of sixteen per lane, through
    gl_debug_field_op 100-109. Inside a transform only the first stage of the first pass sees operands that make add's second wrap or
    sub's second borrow fire; here every stage does: the vectors mix 0, tiny values, values just below 2^64 (>= p) and random words, so
    the correction blocks — also the constants e 2^K that settle a pending correction behind a shift, and the role-swapped butterflies
    of the signed radix 4 — run in most wavefronts. (Checked once with two deliberately wrong builds: a correction constant off by
    one fails the stage comparisons, sum and difference roles exchanged in the signed radix 4's correction fail the last ones.)"""
    rng = np.random.default_rng(23)
    n_vec = 4096
    kinds = rng.integers(0, 5, size=(n_vec, 16))
    rnd = rng.integers(0, 2**64, size=(n_vec, 16), dtype=np.uint64)
    small = rng.integers(0, 4, size=(n_vec, 16), dtype=np.uint64)
    top = np.uint64((1 << 64) - 1) - small
    nearp = np.uint64(P) + small
    vec = np.where(kinds == 0, rnd, np.where(kinds == 1, small, np.where(kinds == 2, top, np.where(kinds == 3, nearp, np.uint64(0)))))
    vec[:64] = np.uint64((1 << 64) - 1)   # a whole wavefront of the extreme operand
    vec[64:128, 0::2] = 0                 # 0 - (2^64 - 1) in every butterfly of every stage order
    vec[64:128, 1::2] = np.uint64((1 << 64) - 1)
    vec[128:192, :8] = np.uint64((1 << 64) - 1)
    vec[128:192, 8:] = 0
    flat = np.ascontiguousarray(vec.reshape(-1))
    rows = [[int(x) for x in r] for r in vec.tolist()]

    def run(which):
        return np.array(run_op(gpu, 100 + which, flat), dtype=np.uint64).reshape(n_vec, 16).tolist()

    for s in range(4):
        assert run(s) == [_dif_stage(r, s) for r in rows], ("stage", s)
    def dif16(r):
        for s in (3, 2, 1, 0):
            r = _dif_stage(r, s)
        return r
    assert run(4) == [dif16(r) for r in rows]
    def dif4_blocks(r):
        return _dif_stage(_dif_stage(r, 1, d=2), 0, d=2)
    assert run(5) == [dif4_blocks(r) for r in rows]
    for kbhi in range(4):
        def twiddled(r):
            return dif4_blocks([x * pow(2, 39 * (i & 3) * (4 * kbhi + (i >> 2)), P) % P for i, x in enumerate(r)])
        assert run(6 + kbhi) == [twiddled(r) for r in rows], ("shift twiddles + radix 4", kbhi)
