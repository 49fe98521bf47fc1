"""Device FRI opening pipeline vs oracle/fri_ref.py (itself pinned by prove -> verify): the proof
object is identical element for element, and the oracle's verifier accepts it."""
import numpy as np
import pytest

from fri_instance import make_fri_instance, transcript_before_fri
from gpu_util import gpu  # noqa: F401
from oracle import fri_ref, pyref

pytestmark = pytest.mark.gpu
P = pyref.P


def device_oracles(gpu, oracles, params):
    import plonky2_gpu_amd as pg

    out = []
    for o in oracles:
        b = pg.PolynomialBatch.from_values(gpu, np.array(o["values"], dtype=np.uint64), params["rate_bits"], False, params["cap_height"])
        assert b.merkle_tree.cap.tolist() == o["cap"]
        out.append(b)
    return out


def device_transcript(gpu, oracles, openings):
    import plonky2_gpu_amd as pg

    ch = pg.Challenger(gpu)
    for o in oracles:
        ch.observe_cap(o["cap"])
    for vals in openings:
        ch.observe_extension_elements(vals)
    return ch


def test_challenger_matches_oracle(gpu):
    import plonky2_gpu_amd as pg

    a, b = pg.Challenger(gpu), fri_ref.Challenger()
    for c in (a, b):
        c.observe_elements([1, 2, 3])
    assert a.get_n_challenges(3) == b.get_n_challenges(3)
    for c in (a, b):
        c.observe_elements(list(range(20)))
        c.observe_extension_elements([(5, 6)])
    assert a.get_extension_challenge() == b.get_extension_challenge()
    assert a.get_n_challenges(9) == b.get_n_challenges(9)


@pytest.mark.parametrize("degree_bits,arity_bits,polys", [(4, (2, 1), (3, 2)), (5, (3,), (3, 2)), (4, (1, 1, 1), (2, 1)), (3, (), (3, 2)),
                                                          (10, (4, 3), (5, 3))])
def test_prove_openings_equals_oracle_and_verifies(gpu, degree_bits, arity_bits, polys):
    import plonky2_gpu_amd as pg

    oracles, instance, params, openings = make_fri_instance(degree_bits=degree_bits, arity_bits=arity_bits, polys_per_oracle=polys,
                                                            seed=degree_bits + len(arity_bits))
    batches = device_oracles(gpu, oracles, params)
    # openings computed on the device equal the oracle's
    for (pt, sel), exp in zip(instance["batches"], openings):
        for (oi, pi), e in zip(sel, exp):
            got = batches[oi].eval_polynomials_ext2([pt])[0, pi]
            assert (int(got[0]), int(got[1])) == e
    proof = pg.prove_openings(gpu, instance, batches, device_transcript(gpu, oracles, openings), params)
    exp = fri_ref.prove_openings(instance, oracles, transcript_before_fri(oracles, openings), params)
    assert proof["pow_witness"] == exp["pow_witness"]
    assert proof["final_poly"] == exp["final_poly"]
    assert proof["commit_phase_merkle_caps"] == exp["commit_phase_merkle_caps"]
    assert proof["query_round_proofs"] == exp["query_round_proofs"]
    chal = fri_ref.fri_challenges(transcript_before_fri(oracles, openings), proof, degree_bits, params)
    assert fri_ref.verify_fri_proof(instance, openings, chal, [o["cap"] for o in oracles], proof, degree_bits, params)


def test_proof_of_work_is_the_smallest_witness(gpu):
    import ctypes

    from plonky2_gpu_amd import _lib

    state = np.arange(100, 112, dtype=np.uint64)
    for bits in (6, 12):
        min_lz = bits + (64 - P.bit_length())
        w = ctypes.c_uint64()
        _lib.call("gl_fri_proof_of_work", state, 3, min_lz, ctypes.addressof(w), gpu.ptr)
        def ok(c):
            s = [int(v) for v in state]
            s[3] = c
            r = pyref.poseidon(s)[7]
            return 64 - r.bit_length() >= min_lz
        assert ok(w.value)
        lo = max(0, w.value - 300)
        assert not any(ok(c) for c in range(lo, w.value))


def _rand_ext(rng, n):
    return [(rng.randrange(P), rng.randrange(P)) for _ in range(n)]


def _planar(ext):
    return np.array([[e[0] for e in ext], [e[1] for e in ext]], dtype=np.uint64).reshape(-1)


def _unplanar(a):
    a = a.reshape(2, -1)
    return [(int(x), int(y)) for x, y in zip(a[0], a[1])]


@pytest.mark.parametrize("log_n,m", [(0, 1), (3, 1), (4, 5), (9, 3), (12, 37), (15, 2), (18, 3)])
def test_reduce_polys_base(gpu, log_n, m):
    import random

    from plonky2_gpu_amd import _lib
    from plonky2_gpu_amd.device import DeviceBuffer

    rng = random.Random(log_n * 100 + m)
    n = 1 << log_n
    polys = [[rng.randrange(P) for _ in range(n)] for _ in range(m)]
    alpha = (rng.randrange(P), rng.randrange(P))
    d_polys = DeviceBuffer.from_host(gpu, np.array(polys, dtype=np.uint64).reshape(-1))
    ptrs = np.array([d_polys.ptr + 8 * n * j for j in range(m)], dtype=np.uint64)
    d_ptrs = DeviceBuffer.from_host(gpu, ptrs)
    d_out = DeviceBuffer(gpu, 2 * n)
    _lib.call("gl_fri_reduce_polys_base", d_ptrs.ptr, m, n, np.array(alpha, dtype=np.uint64), d_out.ptr, gpu.ptr)
    assert _unplanar(d_out.download()) == fri_ref.reduce_polys_base(polys, alpha)


@pytest.mark.parametrize("log_n", [1, 2, 5, 8, 9, 13, 16, 18, 20])  # 2^18: the ed25519 proof; 2^20: config #3's degree
def test_divide_by_linear_accumulate(gpu, log_n):
    import random

    from plonky2_gpu_amd import _lib
    from plonky2_gpu_amd.device import DeviceBuffer

    rng = random.Random(log_n)
    n = 1 << log_n
    comp1, comp2 = _rand_ext(rng, n), _rand_ext(rng, n)
    z1, z2, scale = _rand_ext(rng, 3)
    d_final = DeviceBuffer(gpu, 2 * n)
    for comp, z, acc in ((comp1, z1, 0), (comp2, z2, 1)):
        d_comp = DeviceBuffer.from_host(gpu, _planar(comp))
        _lib.call("gl_fri_divide_by_linear", d_comp.ptr, n, np.array(z, dtype=np.uint64),
                  np.array(scale, dtype=np.uint64), acc, d_final.ptr, gpu.ptr)
    q1, q2 = fri_ref.divide_by_linear(comp1, z1), fri_ref.divide_by_linear(comp2, z2)
    exp = [(0, 0)] + [fri_ref.ext_add(fri_ref.ext_mul(a, scale), b) for a, b in zip(q1, q2)]
    assert _unplanar(d_final.download()) == exp


@pytest.mark.parametrize("log_len,ab", [(1, 1), (4, 2), (6, 3), (10, 4), (14, 1), (14, 4), (18, 4), (19, 2)])
def test_fold_and_interleave(gpu, log_len, ab):
    import random

    from plonky2_gpu_amd import _lib
    from plonky2_gpu_amd.device import DeviceBuffer

    rng = random.Random(log_len + ab)
    n = 1 << log_len
    coeffs = _rand_ext(rng, n)
    beta = (rng.randrange(P), rng.randrange(P))
    d_c = DeviceBuffer.from_host(gpu, _planar(coeffs))
    d_o = DeviceBuffer(gpu, 2 * (n >> ab))
    _lib.call("gl_fri_fold", d_c.ptr, n, ab, np.array(beta, dtype=np.uint64), d_o.ptr, gpu.ptr)
    assert _unplanar(d_o.download()) == [fri_ref.reduce_with_powers_ext(coeffs[k : k + (1 << ab)], beta) for k in range(0, n, 1 << ab)]
    d_r = DeviceBuffer(gpu, 2 * n)
    _lib.call("gl_ext2_interleave", d_c.ptr, n, d_r.ptr, gpu.ptr)
    assert d_r.download().tolist() == fri_ref.flatten(coeffs)
