#!/usr/bin/env python3
"""Generate tests/golden/prove_full_2e13.bin: the proof the oracle prover (oracle/prove_ref.py, with the C restatement for
NTT / Poseidon / Merkle, oracle/accel.py) gives for the 13-gate circuit of tests/plonk_instance.make_full_circuit at 2^13
rows — 135 wires, LDE 2^16, FRI arities [4, 4, 4], cap height 4, 28 query rounds. It takes the oracle ~4 minutes, too long for
the GPU suite, so the bytes are a fixture; the circuit and the witness are rebuilt from the seed by the test
(tests/test_gpu_prove.py::test_proof_bytes_at_2e13_rows_equal_the_fixture). At this size the wires commitment inside gl_prove
takes the pipelined branch (48+ columns, 2^16 leaves). Run:  python tests/golden/gen_prove_golden.py
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
from plonk_instance import make_full_circuit  # noqa: E402

PARAMS = dict(degree_bits=13, seed=5, arity_bits=(4, 4, 4), cap_height=4, num_queries=28)


def main():
    t0 = time.time()
    with accel.c_backend():
        circuit, wires, pis = make_full_circuit(PARAMS["degree_bits"], seed=PARAMS["seed"], arity_bits=PARAMS["arity_bits"],
                                                cap_height=PARAMS["cap_height"], num_queries=PARAMS["num_queries"])
        proof = prove_ref.prove(circuit, wires, pis)
        assert prove_ref.verify(circuit, proof)
    data = serialize_ref.proof_bytes(proof)
    out = os.path.join(ROOT, "tests", "golden", "prove_full_2e13.bin")
    with open(out, "wb") as f:
        f.write(data)
    meta = dict(PARAMS, arity_bits=list(PARAMS["arity_bits"]), bytes=len(data), sha256=hashlib.sha256(data).hexdigest(),
                circuit_digest=[int(v) for v in circuit["circuit_digest"]], seconds=round(time.time() - t0, 1))
    with open(os.path.join(ROOT, "tests", "golden", "prove_full_2e13.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print(meta)


if __name__ == "__main__":
    main()
