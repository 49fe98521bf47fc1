"""A circuit with the plonky2-ed25519 circuit's shape and gate table (234 wires / 80 routed, 8 constants, the 25 gates
and 6 selector groups of plonky2_gpu_amd/ed25519_circuit.py with their real parameters) whose rows USE all 25 gate
kinds: row r carries gate r mod 25, filled by the witness generators of oracle/gates_ref.py (restatements of the
reference's SimpleGenerators, cited there). TEST-SIDE ONLY — it imports the oracle.

To reach 2^18 rows in seconds, every gate kind gets a few honestly generated template rows (with their own gate
constants) which are then laid out over the trace with numpy. Rows built from the same template hold identical values,
so the routed cells of one template class can be tied together by copy constraints: sigma sends each such cell to the
same column of the next row of its class (a cycle per class and column) — a non-trivial permutation argument with
real wire equalities, not the identity.
"""
import os
import random
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import synth_circuit as sc  # noqa: E402  numpy field helpers (np_mul, subgroup), FRI arities
from oracle import gates_ref, pyref  # noqa: E402
from plonky2_gpu_amd import ed25519_circuit as ed  # noqa: E402

P = pyref.P


def make_all_gates_circuit(degree_bits, seed=1, templates=4, fri_params=None, pih_of=pyref.hash_no_pad):
    """Returns (circuit, wires [234][n] uint64, public_inputs). `pih_of` hashes the public inputs (the device's sponge in
    the big GPU test, the model's in the CPU ones — the same function)."""
    rng = random.Random(seed * 7907)
    nrng = np.random.default_rng(seed)
    n = 1 << degree_bits
    nw, nr, ncst = ed.NUM_WIRES, ed.NUM_ROUTED_WIRES, ed.NUM_CONSTANTS
    ngates, nsel = len(ed.GATES), len(ed.GROUPS)
    public_inputs = [rng.randrange(P) for _ in range(3)]
    pih = [int(x) for x in pih_of(public_inputs)]
    # gate of every row: cycle through the table; row 0 is the PublicInputGate (plonky2 puts it first)
    row_gate = (np.arange(n, dtype=np.int64) + 2) % ngates
    row_tmpl = (np.arange(n, dtype=np.int64) // ngates) % templates
    wires = sc.np_random(nrng, (nw, n))  # wires a gate does not use are unconstrained
    constants = np.empty((ncst, n), dtype=np.uint64)
    for g in range(nsel):  # selector polynomial of group g: the gate's index inside, UNUSED outside (selectors.rs)
        lo, hi = ed.GROUPS[g]
        constants[g] = np.where((row_gate >= lo) & (row_gate < hi), row_gate, sc.UNUSED_SELECTOR).astype(np.uint64)
    sub = sc.subgroup(degree_bits)
    k_is = [pow(7, j, P) for j in range(nr)]
    sigmas = np.stack([sc.np_mul(sub, np.uint64(k)) for k in k_is])  # identity permutation to start from
    k_arr = np.array(k_is, dtype=np.uint64)
    used_wires = {}
    for g, (kind, param) in enumerate(ed.GATES):
        width = gates_ref.num_wires(kind, param)
        assert width <= nw
        used_wires[g] = width
        for t in range(templates):
            rows = np.flatnonzero((row_gate == g) & (row_tmpl == t))
            if rows.size == 0:
                continue
            c0, c1 = rng.randrange(P), rng.randrange(P)
            row = gates_ref.fill_row(kind, param, rng, [c0, c1], pih)
            assert len(row) == width or kind in ("noop",)
            assert all(v == 0 for v in gates_ref.constraints(kind, param, [c0, c1], row + [0] * (nw - len(row)), pih, gates_ref.Base))
            constants[nsel, rows], constants[nsel + 1, rows] = np.uint64(c0), np.uint64(c1)
            for j, v in enumerate(row):
                wires[j, rows] = np.uint64(v)
            # copy constraints: routed cells of this class in a cycle per column
            if rows.size > 1:
                nxt = np.roll(rows, -1)
                for j in range(min(width, nr)):
                    sigmas[j, rows] = sc.np_mul(np.full(rows.size, k_arr[j], dtype=np.uint64), sub[nxt])
    fp = fri_params or dict(rate_bits=3, cap_height=4, reduction_arity_bits=sc.constant_arity_bits(degree_bits, 3, 4),
                            proof_of_work_bits=16, num_query_rounds=28)
    circuit = dict(degree_bits=degree_bits, num_wires=nw, num_routed_wires=nr, num_constants=ncst, num_challenges=ed.NUM_CHALLENGES,
                   quotient_degree_factor=ed.QUOTIENT_DEGREE_FACTOR, k_is=k_is, constants=constants, sigmas=sigmas,
                   gates=list(ed.GATES), selector_indices=list(ed.SELECTOR_INDICES), groups=list(ed.GROUPS),
                   num_gate_constraints=ed.NUM_GATE_CONSTRAINTS, fri_params=fp)
    return circuit, wires, public_inputs


def as_oracle_circuit(circuit, wires, prove_ref):
    """The same circuit with Python-int columns and its preprocessed commitment, as oracle/prove_ref.py wants it."""
    fp = circuit["fri_params"]
    oc = dict(circuit, constants=[[int(v) for v in c] for c in circuit["constants"]], sigmas=[[int(v) for v in c] for c in circuit["sigmas"]])
    oc["constants_sigmas"] = prove_ref.commit_from_values(oc["constants"] + oc["sigmas"], fp["rate_bits"], fp["cap_height"])
    oc["circuit_digest"] = prove_ref.circuit_digest(oc["constants_sigmas"]["cap"], circuit["degree_bits"])
    return oc, [[int(v) for v in c] for c in wires]
