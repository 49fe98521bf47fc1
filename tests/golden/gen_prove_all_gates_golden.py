#!/usr/bin/env python3
"""Generate tests/golden/prove_all_gates_2e14.bin: the proof the oracle prover (oracle/prove_ref.py, with the C restatement
for NTT / Poseidon / Merkle, oracle/accel.py) gives for the circuit of tests/ed25519_rows.make_all_gates_circuit at 2^14
rows — the 25 gates of the ed25519 circuit with their real parameters, every kind constraining honestly generated rows
(655 rows each), 234 wires, copy constraints, LDE 2^17, FRI arities [4, 4, 4], cap height 4, 28 query rounds. The oracle
needs a quarter of an hour for it, so the bytes are a fixture; circuit and witness are rebuilt from the seed by the test
(tests/test_gpu_prove.py::test_all_25_gates_proof_bytes_at_2e14_rows_equal_the_fixture).
Run:  python tests/golden/gen_prove_all_gates_golden.py
"""
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import accel, prove_ref, serialize_ref  # noqa: E402
import ed25519_rows as er  # noqa: E402

PARAMS = dict(degree_bits=14, seed=6, templates=4,
              fri_params=dict(rate_bits=3, cap_height=4, reduction_arity_bits=[4, 4, 4], proof_of_work_bits=10, num_query_rounds=28))


def main():
    t0 = time.time()
    with accel.c_backend():
        circuit, wires, pis = er.make_all_gates_circuit(PARAMS["degree_bits"], seed=PARAMS["seed"], templates=PARAMS["templates"],
                                                        fri_params=PARAMS["fri_params"])
        oc, ow = er.as_oracle_circuit(circuit, wires, prove_ref)
        proof = prove_ref.prove(oc, ow, pis)
        assert prove_ref.verify(oc, proof)
    data = serialize_ref.proof_bytes(proof)
    with open(os.path.join(ROOT, "tests", "golden", "prove_all_gates_2e14.bin"), "wb") as f:
        f.write(data)
    meta = dict(PARAMS, bytes=len(data), sha256=hashlib.sha256(data).hexdigest(),
                circuit_digest=[int(v) for v in oc["circuit_digest"]], seconds=round(time.time() - t0, 1))
    with open(os.path.join(ROOT, "tests", "golden", "prove_all_gates_2e14.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print(meta)


if __name__ == "__main__":
    main()
