"""Pins oracle/fri_ref.py: the restated FRI verifier (fri/verifier.rs) accepts every proof of the
restated prover (fri/oracle.rs:1047-1112, fri/prover.rs) and rejects corrupted ones — the way the
reference tests itself (prove -> verify)."""
import copy

import pytest

from fri_instance import make_fri_instance, transcript_before_fri
from oracle import fri_ref, pyref

P = pyref.P


@pytest.mark.parametrize("degree_bits,arity_bits", [(4, (2, 1)), (5, (3,)), (4, (1, 1, 1)), (3, ())])
def test_prove_then_verify(degree_bits, arity_bits):
    oracles, instance, params, openings = make_fri_instance(degree_bits=degree_bits, arity_bits=arity_bits, seed=degree_bits)
    ch = transcript_before_fri(oracles, openings)
    proof = fri_ref.prove_openings(instance, oracles, ch.clone(), params)
    assert len(proof["final_poly"]) == (1 << degree_bits) >> sum(arity_bits)
    chal = fri_ref.fri_challenges(ch.clone(), proof, degree_bits, params)
    caps = [o["cap"] for o in oracles]
    assert fri_ref.verify_fri_proof(instance, openings, chal, caps, proof, degree_bits, params)
    # smallest proof-of-work witness: no smaller candidate passes
    assert proof["pow_witness"] < 64
    # a wrong opening, a wrong final polynomial and a tampered evaluation are rejected
    bad_open = copy.deepcopy(openings)
    bad_open[0][0] = ((bad_open[0][0][0] + 1) % P, bad_open[0][0][1])
    with pytest.raises(AssertionError):
        fri_ref.verify_fri_proof(instance, bad_open, chal, caps, proof, degree_bits, params)
    bad = copy.deepcopy(proof)
    bad["final_poly"][0] = ((bad["final_poly"][0][0] + 1) % P, bad["final_poly"][0][1])
    with pytest.raises(AssertionError):
        fri_ref.verify_fri_proof(instance, openings, chal, caps, bad, degree_bits, params)
    if arity_bits:
        bad = copy.deepcopy(proof)
        e = bad["query_round_proofs"][0]["steps"][0]["evals"]
        e[0] = ((e[0][0] + 1) % P, e[0][1])
        with pytest.raises(AssertionError):
            fri_ref.verify_fri_proof(instance, openings, chal, caps, bad, degree_bits, params)


def test_challenger_duplex_semantics():
    """challenger.rs:43-149: buffered inputs are absorbed in overwrite mode on the next squeeze,
    outputs are popped from the END of the rate portion, an observation invalidates outputs."""
    c = fri_ref.Challenger()
    c.observe_elements([1, 2, 3])
    a = c.get_challenge()
    st = pyref.poseidon([1, 2, 3] + [0] * 9)
    assert a == st[7] and c.get_challenge() == st[6]
    c.observe_element(9)
    st2 = pyref.poseidon([9] + st[1:])
    assert c.get_challenge() == st2[7]
    c2 = fri_ref.Challenger()
    c2.observe_elements(list(range(8)))  # a full rate block duplexes immediately
    assert c2.output_buffer == pyref.poseidon(list(range(8)) + [0] * 4)[:8]


def test_divide_by_linear_and_reduce():
    import random

    rng = random.Random(4)
    coeffs = [(rng.randrange(P), rng.randrange(P)) for _ in range(9)]
    z = (rng.randrange(P), rng.randrange(P))
    q = fri_ref.divide_by_linear(coeffs, z)
    # q(X) (X - z) + p(z) == p(X): compare coefficients
    pz = (0, 0)
    for c in reversed(coeffs):
        pz = fri_ref.ext_add(fri_ref.ext_mul(pz, z), c)
    rebuilt = [fri_ref.ext_sub((0, 0), fri_ref.ext_mul(q[0], z))]
    for i in range(1, len(q)):
        rebuilt.append(fri_ref.ext_sub(q[i - 1], fri_ref.ext_mul(q[i], z)))
    rebuilt.append(q[-1])
    rebuilt[0] = fri_ref.ext_add(rebuilt[0], pz)
    assert rebuilt == coeffs
