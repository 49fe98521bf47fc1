"""Proof wire format (plonky2/src/util/serialization.rs): the product's writer equals the oracle's
independent restatement byte for byte, reading gives the proof back, the parsed proof verifies, and
the byte count matches the closed form implied by the format."""
import pytest

from oracle import prove_ref, serialize_ref
from plonk_instance import make_circuit
from plonky2_gpu_amd import serialization


@pytest.fixture(scope="module")
def proven():
    circuit, wires, pis = make_circuit(4, seed=17, two_groups=True, arity_bits=(2, 1))
    return circuit, prove_ref.prove(circuit, wires, pis)


def test_writer_equals_restatement_and_round_trips(proven):
    circuit, proof = proven
    data = serialization.proof_to_bytes(proof)
    assert data == serialize_ref.proof_bytes(proof)
    back = serialization.proof_from_bytes(data, circuit)
    assert back["openings"] == proof["openings"]
    assert back["wires_cap"] == proof["wires_cap"] and back["public_inputs"] == proof["public_inputs"]
    assert serialization.proof_to_bytes(back) == data
    assert prove_ref.verify(circuit, back)


def test_size_formula(proven):
    circuit, proof = proven
    fp = circuit["fri_params"]
    cap = 32 << fp["cap_height"]
    nch, qdf = circuit["num_challenges"], circuit["quotient_degree_factor"]
    npp = -(-circuit["num_routed_wires"] // qdf) - 1
    leaf_lens = [circuit["num_constants"] + circuit["num_routed_wires"], circuit["num_wires"], nch * (1 + npp), nch * qdf]
    openings = 16 * (circuit["num_constants"] + circuit["num_routed_wires"] + circuit["num_wires"] + 2 * nch + npp * nch + qdf * nch)
    lde_bits = circuit["degree_bits"] + fp["rate_bits"]
    per_round = sum(8 * n + 1 + 32 * (lde_bits - fp["cap_height"]) for n in leaf_lens)
    bits = lde_bits
    for ab in fp["reduction_arity_bits"]:
        bits -= ab
        per_round += 16 * (1 << ab) + 1 + 32 * max(bits - fp["cap_height"], 0)
    final = 16 << (circuit["degree_bits"] - sum(fp["reduction_arity_bits"]))
    expect = 3 * cap + openings + cap * len(fp["reduction_arity_bits"]) + fp["num_query_rounds"] * per_round + final + 8 + 8 * 3
    assert len(serialization.proof_to_bytes(proof)) == expect


def test_truncated_and_non_canonical_input_is_rejected(proven):
    circuit, proof = proven
    data = serialization.proof_to_bytes(proof)
    with pytest.raises(EOFError):
        serialization.proof_from_bytes(data[: len(data) // 2], circuit)
    bad = b"\xff" * 8 + data[8:]
    with pytest.raises(ValueError):
        serialization.proof_from_bytes(bad, circuit)
