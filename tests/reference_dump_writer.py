"""Plays the Rust side of tools/reference_dumps.py: writes the raw-u64 dump files of the reference
(plonky2/src/plonk/circuit_builder.rs:1077-1117, fri/oracle.rs:743-753, plonk/prover.rs:829-877; read by cuda/test.cu:129-136,
412-428) for one proof of a circuit with the ed25519 shape, every value computed by the CPU oracle. TEST-SIDE ONLY.

    python tests/reference_dump_writer.py <dir> [--degree-bits 7]
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def write(directory, degree_bits=7, seed=11, templates=2, fri_params=None):
    from oracle import accel, prove_ref, serialize_ref
    import ed25519_rows as er

    os.makedirs(directory, exist_ok=True)

    def dump(name, data):
        np.ascontiguousarray(np.array(data, dtype=np.uint64)).astype("<u8").tofile(os.path.join(directory, name))

    fp = fri_params or dict(rate_bits=3, cap_height=4, reduction_arity_bits=[2, 1], proof_of_work_bits=5, num_query_rounds=6)
    with accel.c_backend():
        circuit, wires, pis = er.make_all_gates_circuit(degree_bits, seed=seed, templates=templates, fri_params=fp)
        oc, ow = er.as_oracle_circuit(circuit, wires, prove_ref)
        trace = {}
        proof = prove_ref.prove(oc, ow, pis, trace)
        assert prove_ref.verify(oc, proof)
    dump("values.bin", ow)                                            # fri/oracle.rs:743-753
    dump("sigma_vecs.bin", oc["sigmas"])                              # circuit_builder.rs:1097-1099
    for name, c in (("constants_sigmas_commitment", oc["constants_sigmas"]), ("zs_partial_products_commitment", trace["zs_partial_products_commitment"]),
                    ("wires_commitment", trace["wires_commitment"])):
        dump(name + ".polynomials.bin", c["polynomials"])             # circuit_builder.rs:1103-1115, prover.rs:832-847
        dump(name + ".leaves.bin", c["leaves"])
        dump(name + ".digests.bin", c["digests"])
        dump(name + ".caps.bin", c["cap"])
    dump("zs_partial_products.bin", trace["zs_partial_products"])
    for k in ("alphas", "betas", "gammas"):                           # prover.rs:849-860
        dump(k + ".bin", trace[k])
    dump("k_is.bin", oc["k_is"])                                      # prover.rs:861-864
    dump("quotient_values2.bin", trace["quotient_polys"])             # the layout of cuda/test.cu:553-566: [2][n_ext] coefficients
    dump("public_inputs.bin", pis)
    dump("public_inputs_hash.bin", trace["public_inputs_hash"])
    open(os.path.join(directory, "proof.bin"), "wb").write(serialize_ref.proof_bytes(proof))
    json.dump(fp, open(os.path.join(directory, "fri_params.json"), "w"))
    return oc, proof


if __name__ == "__main__":
    import argparse

    ap = argparse.ArgumentParser()
    ap.add_argument("directory")
    ap.add_argument("--degree-bits", type=int, default=7)
    a = ap.parse_args()
    write(a.directory, a.degree_bits)
    print("wrote", sorted(os.listdir(a.directory)))
