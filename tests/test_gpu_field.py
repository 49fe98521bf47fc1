"""gl_field.cuh against Python big ints on the reference's edge operands
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
