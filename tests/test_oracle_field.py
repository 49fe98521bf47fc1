"""Pins the C oracle's field arithmetic against Python big ints on the reference's own edge
operands (field/src/prime_field_testing.rs:7-17, 79-180)."""
import itertools

from oracle import pyref

P = pyref.P


def edge_operands():
    # prime_field_testing.rs:7-17: values near 0, 2^31, 2^32, 2^63 and p, word size 64
    base = list(range(0, 10))
    for c in (1 << 31, 1 << 32, 1 << 63):
        base += list(range(c - 10, c + 11))
    base += list(range(P - 10, P))
    # non-canonical representatives are legal inputs too (goldilocks_field.rs:26)
    base += [P, P + 1, (1 << 64) - 1, (1 << 64) - 2, P + (1 << 31)]
    return sorted(set(base))


def test_binary_ops_match_bigint(oracle):
    L = oracle.lib()
    ops = edge_operands()
    for a, b in itertools.product(ops, ops):
        assert L.glo_canon(L.glo_add(a, b)) == (a + b) % P
        assert L.glo_canon(L.glo_sub(a, b)) == (a - b) % P
        assert L.glo_canon(L.glo_mul(a, b)) == (a * b) % P
    for a in ops:
        assert L.glo_canon(L.glo_neg(a)) == (-a) % P
        assert L.glo_canon(a) == a % P
        assert L.glo_canon(L.glo_mac(a, ops[-1], ops[-2])) == (a + ops[-1] * ops[-2]) % P


def test_inverse_and_identities(oracle):
    L = oracle.lib()
    for a in edge_operands():
        if a % P == 0:
            continue
        inv = L.glo_inverse(a)
        assert inv < P and (inv * a) % P == 1
    # prime_field_testing.rs:146-180
    for e in [0, 1, 2, 3, 4, 30, 31, 32, 33, 34, 3936]:
        assert (L.glo_inverse_2exp(e) * pow(2, e, P)) % P == 1
    assert (((P + 1) // 2) * 2) % P == 1
    assert L.glo_canon(L.glo_sub(0, 1)) == P - 1


def test_constants(oracle):
    L = oracle.lib()
    # goldilocks_field.rs:82,89; cuda/test.cu:195; cuda/plonky2_gpu.cu:746
    assert pow(7, (P - 1) >> 32, P) == 1753635133440165772
    assert L.glo_primitive_root_of_unity(32) == 1753635133440165772
    assert L.glo_inverse_2exp(18) == 18446673700670423041
    assert L.glo_inverse_2exp(21) == 0xFFFFF7FF00000801
    for k in range(1, 33):
        w = L.glo_canon(L.glo_primitive_root_of_unity(k))
        assert w == pyref.root_of_unity(k)
        assert pow(w, 1 << k, P) == 1 and pow(w, 1 << (k - 1), P) == P - 1
    assert L.glo_canon(L.glo_exp(7, P - 1)) == 1
