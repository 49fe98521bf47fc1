"""serialize_ref.py — independent restatement of the proof wire format (TEST INFRASTRUCTURE ONLY),
written as one flat numpy pass over the proof's fields in the order of
Write::write_proof_with_public_inputs (plonky2/src/util/serialization.rs:641-689): every field
element is a canonical little-endian u64, a Merkle proof is prefixed by its length as one byte.
The reference holds no byte fixtures for proofs; this pins the ORDER and the framing, the
round trip through the product's reader pins the shapes (tests/test_serialization.py)."""
import numpy as np

P = 0xFFFFFFFF00000001


def _u64(xs):
    return np.array([int(x) % P for x in xs], dtype="<u8").tobytes()


def _flat_ext(v):
    return [c for e in v for c in e]


def proof_bytes(proof):
    out = []
    for cap in (proof["wires_cap"], proof["plonk_zs_partial_products_cap"], proof["quotient_polys_cap"]):
        out.append(_u64([x for h in cap for x in h]))
    op = proof["openings"]
    for k in ("constants", "plonk_sigmas", "wires", "plonk_zs", "plonk_zs_next", "partial_products", "quotient_polys"):
        out.append(_u64(_flat_ext(op[k])))
    fp = proof["opening_proof"]
    for cap in fp["commit_phase_merkle_caps"]:
        out.append(_u64([x for h in cap for x in h]))
    for rnd in fp["query_round_proofs"]:
        for evals, sib in rnd["initial_trees_proof"]:
            out += [_u64(evals), bytes([len(sib)]), _u64([x for h in sib for x in h])]
        for st in rnd["steps"]:
            out += [_u64(_flat_ext(st["evals"])), bytes([len(st["merkle_proof"])]), _u64([x for h in st["merkle_proof"] for x in h])]
    out += [_u64(_flat_ext(fp["final_poly"])), _u64([fp["pow_witness"]]), _u64(proof["public_inputs"])]
    return b"".join(out)
