"""Pins oracle/prove_ref.py the way the reference pins its prover: prove -> verify, and the verifier
rejects anything tampered with (plonky2/src/plonk/prover.rs + verifier.rs)."""
import copy

import pytest

from oracle import prove_ref, pyref
from plonk_instance import make_circuit

P = pyref.P


@pytest.mark.parametrize("degree_bits,two_groups,arity_bits", [(4, False, (2, 1)), (4, True, (1, 2)), (5, True, (3,))])
def test_prove_then_verify(degree_bits, two_groups, arity_bits):
    circuit, wires, pis = make_circuit(degree_bits, seed=3 + degree_bits, two_groups=two_groups, arity_bits=arity_bits)
    proof = prove_ref.prove(circuit, wires, pis)
    assert prove_ref.verify(circuit, proof)


def test_verifier_rejects_tampering():
    circuit, wires, pis = make_circuit(4, seed=11)
    proof = prove_ref.prove(circuit, wires, pis)
    assert prove_ref.verify(circuit, proof)

    def bump(pair):
        return ((pair[0] + 1) % P, pair[1])

    for mutate in (
        lambda p: p["openings"]["wires"].__setitem__(3, bump(p["openings"]["wires"][3])),
        lambda p: p["openings"]["quotient_polys"].__setitem__(0, bump(p["openings"]["quotient_polys"][0])),
        lambda p: p["openings"]["plonk_zs_next"].__setitem__(1, bump(p["openings"]["plonk_zs_next"][1])),
        lambda p: p["opening_proof"]["final_poly"].__setitem__(0, bump(p["opening_proof"]["final_poly"][0])),
        lambda p: p["public_inputs"].__setitem__(0, (p["public_inputs"][0] + 1) % P),
        lambda p: p["wires_cap"][0].__setitem__(0, (p["wires_cap"][0][0] + 1) % P),
        lambda p: p["opening_proof"].__setitem__("pow_witness", p["opening_proof"]["pow_witness"] + 1),
    ):
        bad = copy.deepcopy(proof)
        mutate(bad)
        with pytest.raises(AssertionError):
            prove_ref.verify(circuit, bad)


def test_unsatisfied_witness_does_not_verify():
    """quotient_degree_factor 8 is a power of two, so nothing is trimmed and the prover itself cannot
    notice (prover.rs:161-165 only checks the trimmed tail); the verifier does."""
    circuit, wires, pis = make_circuit(4, seed=5)
    wires = [list(c) for c in wires]
    wires[3] = [(v + 1) % P for v in wires[3]]  # breaks every arithmetic row's first output
    proof = prove_ref.prove(circuit, wires, pis)
    with pytest.raises(AssertionError):
        prove_ref.verify(circuit, proof)


def test_synthetic_circuit_generator_gives_a_provable_circuit():
    """tools/synth_circuit.py (numpy witness generator used at bench sizes) against the oracle prover."""
    import os
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tools"))
    import synth_circuit

    fp = dict(rate_bits=3, cap_height=1, reduction_arity_bits=[2, 1], proof_of_work_bits=2, num_query_rounds=2)
    circuit, wires, pis = synth_circuit.make(4, num_wires=14, num_routed=12, num_constants=4, seed=3, fri_params=fp)
    synth_circuit.set_public_input_row(wires, pyref.hash_no_pad(pis))
    circuit["constants"] = [[int(v) for v in c] for c in circuit["constants"]]
    circuit["sigmas"] = [[int(v) for v in c] for c in circuit["sigmas"]]
    wires = [[int(v) for v in c] for c in wires]
    circuit["constants_sigmas"] = prove_ref.commit_from_values(circuit["constants"] + circuit["sigmas"], 3, 1)
    circuit["circuit_digest"] = prove_ref.circuit_digest(circuit["constants_sigmas"]["cap"], 4)
    assert prove_ref.verify(circuit, prove_ref.prove(circuit, wires, pis))


def test_synthetic_circuit_with_the_ed25519_gate_table_is_provable():
    """gate_table="ed25519": the circuit declares all 25 gates of the ed25519 table (6 selector groups, 231 constraints)
    and instantiates Noop / Constant / PublicInput / Arithmetic{20} out of it. The oracle prover's quotient must
    divide (every declared gate is evaluated at every LDE point, the unused ones filtered out on the subgroup by
    their selector polynomials) and the oracle verifier, which evaluates all 25 gates at zeta in the extension
    field, must accept; a corrupted arithmetic output must not survive."""
    import os
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tools"))
    import synth_circuit

    from plonky2_gpu_amd import ed25519_circuit as ed

    fp = dict(rate_bits=3, cap_height=1, reduction_arity_bits=[2], proof_of_work_bits=2, num_query_rounds=2)
    circuit, wires, pis = synth_circuit.make(4, num_wires=234, num_routed=80, num_constants=8, seed=6, fri_params=fp, gate_table="ed25519")
    assert circuit["gates"] == ed.GATES and circuit["num_gate_constraints"] == 231 and len(circuit["groups"]) == 6
    used = sorted(set(int(v) for v in circuit["constants"][0]))
    assert set(used) <= {0, 1, 2, 5} and 5 in used
    assert all(int(v) == synth_circuit.UNUSED_SELECTOR for c in circuit["constants"][1:6] for v in c)
    synth_circuit.set_public_input_row(wires, pyref.hash_no_pad(pis))
    circuit["constants"] = [[int(v) for v in c] for c in circuit["constants"]]
    circuit["sigmas"] = [[int(v) for v in c] for c in circuit["sigmas"]]
    wires = [[int(v) for v in c] for c in wires]
    circuit["constants_sigmas"] = prove_ref.commit_from_values(circuit["constants"] + circuit["sigmas"], 3, 1)
    circuit["circuit_digest"] = prove_ref.circuit_digest(circuit["constants_sigmas"]["cap"], 4)
    assert prove_ref.verify(circuit, prove_ref.prove(circuit, wires, pis))
    row = circuit["constants"][0].index(5)  # an arithmetic row: break its first output
    bad = [list(c) for c in wires]
    bad[3][row] = (bad[3][row] + 1) % P
    with pytest.raises(AssertionError):
        prove_ref.verify(circuit, prove_ref.prove(circuit, bad, pis))


def test_full_gate_list_prove_then_verify():
    """every gate kind of the ed25519 gate list in one circuit (tests/plonk_instance.make_full_circuit)"""
    from plonk_instance import make_full_circuit

    circuit, wires, pis = make_full_circuit(4, seed=2)
    proof = prove_ref.prove(circuit, wires, pis)
    assert prove_ref.verify(circuit, proof)
    bad = [list(c) for c in wires]
    bad[30] = [(v + 1) % P for v in bad[30]]
    with pytest.raises(AssertionError):
        prove_ref.verify(circuit, prove_ref.prove(circuit, bad, pis))


@pytest.mark.parametrize("qdf,two_groups", [(5, False), (6, False), (7, True), (4, True)])
def test_quotient_degree_factor_that_is_not_a_power_of_two(qdf, two_groups):
    """prover.rs:153-166: the quotient is trimmed to quotient_degree_factor * n coefficients (the tail must
    be zero for a satisfied circuit) and split into that many degree-n chunks."""
    circuit, wires, pis = make_circuit(4, seed=20 + qdf, two_groups=two_groups, quotient_degree_factor=qdf)
    proof = prove_ref.prove(circuit, wires, pis)
    assert len(proof["openings"]["quotient_polys"]) == 2 * qdf
    assert prove_ref.verify(circuit, proof)


@pytest.mark.parametrize("num_challenges", [1, 3])
def test_other_numbers_of_challenges(num_challenges):
    """config.num_challenges sizes the Z / partial-product / quotient batches and the alpha reduction
    (prover.rs:97-151); 2 is only the standard configuration's value."""
    circuit, wires, pis = make_circuit(4, seed=40 + num_challenges, num_challenges=num_challenges)
    proof = prove_ref.prove(circuit, wires, pis)
    assert len(proof["openings"]["plonk_zs"]) == num_challenges
    assert len(proof["openings"]["quotient_polys"]) == 8 * num_challenges
    assert prove_ref.verify(circuit, proof)


@pytest.mark.parametrize("which", ["mini", "mini2", "full"])
def test_c_backend_gives_the_same_proof(which):
    """oracle/accel.py serves Poseidon / Merkle / NTT from the pinned C restatement so that prove_ref reaches 2^10..2^12
    rows in the GPU parity tests. Same circuits, with and without it: the circuit digest, the proof and the verifier's
    verdict must not change."""
    from oracle import accel
    from plonk_instance import make_full_circuit

    def build():
        if which == "full":
            return make_full_circuit(4, seed=7)
        return make_circuit(5 if which == "mini" else 4, seed=31, two_groups=which == "mini2", arity_bits=(2, 1))

    circuit, wires, pis = build()
    pure = prove_ref.prove(circuit, wires, pis)
    with accel.c_backend():
        c2, w2, p2 = build()
        fast = prove_ref.prove(circuit, wires, pis)
        assert prove_ref.verify(circuit, fast)
    assert c2["circuit_digest"] == circuit["circuit_digest"] and c2["constants_sigmas"]["digests"] == circuit["constants_sigmas"]["digests"]
    assert w2 == wires and p2 == pis
    assert fast == pure
    from oracle import pyref
    assert pyref.poseidon.__module__ == "oracle.pyref"  # the backend is gone when the block ends


def test_all_25_ed25519_gates_with_honest_rows_prove_then_verify():
    """tests/ed25519_rows.py: the ed25519 gate table with its real parameters, every gate kind instantiated by the witness
    generators of oracle/gates_ref.py, copy constraints between equal cells. The oracle proves it and its verifier
    accepts; a single corrupted limb of a U32 gate row is caught."""
    from oracle import accel
    import ed25519_rows as er

    fp = dict(rate_bits=3, cap_height=1, reduction_arity_bits=[2, 1], proof_of_work_bits=2, num_query_rounds=2)
    with accel.c_backend():
        circuit, wires, pis = er.make_all_gates_circuit(6, seed=3, templates=2, fri_params=fp)
        oc, ow = er.as_oracle_circuit(circuit, wires, prove_ref)
        proof = prove_ref.prove(oc, ow, pis)
        assert prove_ref.verify(oc, proof)
        assert sorted(set((r + 2) % 25 for r in range(64))) == list(range(25))  # every gate kind has rows
        bad = [list(c) for c in ow]
        row = next(r for r in range(64) if (r + 2) % 25 == 18)  # a U32ArithmeticGate{6} row
        bad[40][row] = (bad[40][row] + 1) % prove_ref.P
        with pytest.raises(AssertionError):
            prove_ref.verify(oc, prove_ref.prove(oc, bad, pis))


def test_the_2e13_row_proof_fixture_is_what_its_generator_says():
    """tests/golden/prove_full_2e13.bin (tests/golden/gen_prove_golden.py; compared with gl_prove's bytes on the GPU): the file has the
    recorded hash, parses in the proof's wire format for the circuit rebuilt from the recorded seed — whose digest is the recorded
    one — and the oracle's verifier accepts it."""
    import hashlib
    import json
    import os

    from oracle import accel, serialize_ref
    from plonk_instance import make_full_circuit

    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    meta = json.load(open(os.path.join(gold, "prove_full_2e13.json")))
    data = open(os.path.join(gold, "prove_full_2e13.bin"), "rb").read()
    assert hashlib.sha256(data).hexdigest() == meta["sha256"] and len(data) == meta["bytes"]
    with accel.c_backend():
        circuit, _, pis = make_full_circuit(meta["degree_bits"], seed=meta["seed"], arity_bits=tuple(meta["arity_bits"]),
                                            cap_height=meta["cap_height"], num_queries=meta["num_queries"])
        assert [int(v) for v in circuit["circuit_digest"]] == meta["circuit_digest"]
        from plonky2_gpu_amd import serialization  # host-side reader of the wire format (no device involved)

        proof = serialization.proof_from_bytes(data, circuit)
        assert [int(v) for v in proof["public_inputs"]] == pis
        assert serialize_ref.proof_bytes(proof) == data
        assert prove_ref.verify(circuit, proof)


def test_the_2e14_row_all_gates_proof_fixture_is_what_its_generator_says():
    """tests/golden/prove_all_gates_2e14.bin (tests/golden/gen_prove_all_gates_golden.py; compared with gl_prove's bytes on the
    GPU): recorded hash, wire format of the circuit rebuilt from the recorded seed (whose digest is the recorded one), and the
    oracle's verifier — which evaluates all 25 gates of the ed25519 table at zeta over F_p^2 on its own — accepts it."""
    import hashlib
    import json
    import os

    import ed25519_rows as er
    from oracle import accel, serialize_ref

    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    meta = json.load(open(os.path.join(gold, "prove_all_gates_2e14.json")))
    data = open(os.path.join(gold, "prove_all_gates_2e14.bin"), "rb").read()
    assert hashlib.sha256(data).hexdigest() == meta["sha256"] and len(data) == meta["bytes"]
    with accel.c_backend():
        circuit, wires, pis = er.make_all_gates_circuit(meta["degree_bits"], seed=meta["seed"], templates=meta["templates"],
                                                        fri_params=meta["fri_params"])
        oc, _ = er.as_oracle_circuit(circuit, wires[:1], prove_ref)
        assert [int(v) for v in oc["circuit_digest"]] == meta["circuit_digest"]
        from plonky2_gpu_amd import serialization  # host-side reader of the wire format (no device involved)

        proof = serialization.proof_from_bytes(data, circuit)
        assert [int(v) for v in proof["public_inputs"]] == pis
        assert serialize_ref.proof_bytes(proof) == data
        assert prove_ref.verify(oc, proof)



def _salts(circuit, seed):
    import random

    rng = random.Random(seed)
    n_ext = 1 << (circuit["degree_bits"] + circuit["fri_params"]["rate_bits"])
    return [[[rng.randrange(P) for _ in range(n_ext)] for _ in range(prove_ref.SALT_SIZE)] for _ in range(3)]


@pytest.mark.parametrize("two_groups", [False, True])
def test_blinded_proof_verifies_and_differs_only_through_the_salts(two_groups):
    """CircuitConfig::zero_knowledge (plonk/circuit_data.rs:74): the wires, Zs / partial products and quotient commitments get
    SALT_SIZE = 4 random elements per leaf (prover.rs:84, 125, 174; fri/oracle.rs:985-1002), the proof's initial-tree openings
    carry them, the verifier strips them (fri/proof.rs:45-52). prove -> verify, serialise -> parse, other salts -> other caps,
    the same openings at zeta only where the transcript allows (it does not: the caps feed the challenges)."""
    from oracle import serialize_ref
    from plonky2_gpu_amd import serialization

    circuit, wires, pis = make_circuit(4, seed=77, two_groups=two_groups)
    circuit = dict(circuit, fri_params=dict(circuit["fri_params"], hiding=True))
    salts = _salts(circuit, 1)
    proof = prove_ref.prove(circuit, wires, pis, salts=salts)
    assert prove_ref.verify(circuit, proof)
    nw = circuit["num_wires"]
    for rnd in proof["opening_proof"]["query_round_proofs"]:
        lens = [len(evals) for evals, _ in rnd["initial_trees_proof"]]
        assert lens[0] == circuit["num_constants"] + circuit["num_routed_wires"] and lens[1] == nw + 4  # constants/sigmas are never blinded
    data = serialize_ref.proof_bytes(proof)
    parsed = serialization.proof_from_bytes(data, circuit)
    assert serialization.proof_to_bytes(parsed) == data and prove_ref.verify(circuit, parsed)
    other = prove_ref.prove(circuit, wires, pis, salts=_salts(circuit, 2))
    assert other["wires_cap"] != proof["wires_cap"] and prove_ref.verify(circuit, other)
    # a salt element changed in an opened leaf breaks its Merkle proof
    bad = serialization.proof_from_bytes(data, circuit)
    evals, sib = bad["opening_proof"]["query_round_proofs"][0]["initial_trees_proof"][1]
    evals[-1] = (evals[-1] + 1) % P
    with pytest.raises(AssertionError):
        prove_ref.verify(circuit, bad)
    # the flag and the salts go together
    with pytest.raises(AssertionError):
        prove_ref.prove(circuit, wires, pis)
