"""oracle/prove_oracle.c (the C restatement of prove() above the commit: gates, permutation argument, quotient, openings,
FRI, wire format) held against the independent Python restatement oracle/prove_ref.py + serialize_ref.py: identical proof
BYTES on every small circuit shape the Python prover is used for, identical intermediate objects (challenges, Zs / partial
products, quotient polynomials), identical gate constraints gate by gate, and the committed golden proofs (2^13 and 2^14 rows,
written by the Python prover). The C prover is what the GPU suite compares full-size proofs with (tests/test_gpu_prove.py)."""
import hashlib
import json
import os
import random

import numpy as np
import pytest

from oracle import accel, gates_ref, prove_c, prove_ref, pyref, serialize_ref
from plonk_instance import FULL_GATES, make_circuit, make_full_circuit

P = pyref.P
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("kwargs", [
    dict(degree_bits=4, seed=3), dict(degree_bits=5, seed=4, two_groups=True, arity_bits=(3,)), dict(degree_bits=4, seed=9, quotient_degree_factor=5),
    dict(degree_bits=4, seed=10, quotient_degree_factor=6), dict(degree_bits=4, seed=8, two_groups=True, quotient_degree_factor=4),
    dict(degree_bits=4, seed=11, num_challenges=3), dict(degree_bits=4, seed=13, num_challenges=1), dict(degree_bits=3, seed=12, arity_bits=()),
    dict(degree_bits=4, seed=14, arity_bits=(1, 1, 1)), dict(degree_bits=6, seed=15, arity_bits=(4,), cap_height=3, num_queries=5, pow_bits=6)])
def test_c_prover_gives_the_python_provers_bytes(kwargs):
    circuit, wires, pis = make_circuit(**kwargs)
    trace_py, trace_c = {}, {}
    exp = prove_ref.prove(circuit, wires, pis, trace=trace_py)
    c = prove_c.Circuit(circuit, threads=4)
    assert c.circuit_digest == circuit["circuit_digest"]  # circuit_builder.rs:915-927
    assert c.constants_sigmas_cap == circuit["constants_sigmas"]["cap"]
    got = c.prove(wires, pis, trace=trace_c)
    assert got == serialize_ref.proof_bytes(exp)
    for k in ("betas", "gammas", "alphas"):
        assert trace_c[k].tolist() == trace_py[k], k
    assert trace_c["zs_partial_products"].tolist() == trace_py["zs_partial_products"]
    assert trace_c["quotient_polys"].tolist() == trace_py["quotient_polys"]
    assert trace_c["wires_cap"].tolist() == exp["wires_cap"] and trace_c["quotient_cap"].tolist() == exp["quotient_polys_cap"]
    c.close()


def test_c_prover_on_the_circuit_with_every_ed25519_gate_kind():
    with accel.c_backend():
        circuit, wires, pis = make_full_circuit(4, seed=2)
        exp = serialize_ref.proof_bytes(prove_ref.prove(circuit, wires, pis))
    assert prove_c.prove(circuit, wires, pis, threads=4) == exp


def test_recursion_shaped_circuit_with_the_upstream_gate_kinds():
    """A circuit of standard_recursion_config's shape (135 wires, 80 routed, 15 gates in 4 selector groups) whose rows use the eight
    gate kinds of upstream plonky2 beyond the ed25519 list — ArithmeticExtension, MulExtension, Reducing, ReducingExtension,
    Exponentiation, PoseidonMds, LowDegreeInterpolation, HighDegreeInterpolation — next to the basic ones: the Python prover's proof is
    accepted by the verifier (which evaluates every gate over F_p^2 on its own, with the D = 2 extension algebra for the extension
    gates), and the C prover gives the same bytes."""
    from plonk_instance import RECURSION_GATES, make_recursion_circuit

    with accel.c_backend():
        circuit, wires, pis = make_recursion_circuit(5, seed=3)
        assert [k for k, _ in circuit["gates"]] == [k for k, _ in RECURSION_GATES] and circuit["num_gate_constraints"] == 123
        exp = prove_ref.prove(circuit, wires, pis)
        assert prove_ref.verify(circuit, exp)
        bad = [list(c) for c in wires]
        row = next(r for r in range(32) if circuit["constants"][1][r] == 6)  # a LowDegreeInterpolationGate row (selector group 1)
        bad[40][row] = (bad[40][row] + 1) % P
        with pytest.raises(AssertionError):
            prove_ref.verify(circuit, prove_ref.prove(circuit, bad, pis))
    assert prove_c.prove(circuit, wires, pis, threads=4) == serialize_ref.proof_bytes(exp)


def test_c_prover_blinded():
    circuit, wires, pis = make_circuit(5, seed=14, arity_bits=(2, 1))
    circuit = dict(circuit, fri_params=dict(circuit["fri_params"], hiding=True))
    salts = np.random.default_rng(99).integers(0, P, size=(3, 4, 1 << 8), dtype=np.uint64)
    exp = serialize_ref.proof_bytes(prove_ref.prove(circuit, wires, pis, salts=salts.tolist()))
    assert prove_c.prove(circuit, wires, pis, salts=salts, threads=4) == exp
    with pytest.raises(AssertionError, match="hiding"):
        prove_c.prove(circuit, wires, pis, threads=2)


def test_c_prover_reports_a_quotient_that_is_not_a_polynomial():
    circuit, wires, pis = make_circuit(4, seed=25, quotient_degree_factor=5)
    bad = [list(c) for c in wires]
    bad[3] = [(v + 1) % P for v in bad[3]]
    with pytest.raises(AssertionError, match="Quotient has failed"):
        prove_c.prove(circuit, bad, pis, threads=2)


@pytest.mark.parametrize("kind,param", FULL_GATES + [("base_sum", (2, 63)), ("comparison", (32, 16)), ("u32_add_many", (0, 11)), ("u32_add_many", (16, 4)),
                                                       ("u32_range_check", 0), ("u32_range_check", 8), ("random_access", (4, 4, 2)),
                                                       ("u32_subtraction", 11), ("u32_arithmetic", 6), ("arithmetic", 20),
                                                       ("arithmetic_extension", 10), ("mul_extension", 13), ("reducing", 43), ("reducing", 1),
                                                       ("reducing_extension", 32), ("exponentiation", 66), ("exponentiation", 1), ("poseidon_mds", None),
                                                       ("low_degree_interpolation", 4), ("low_degree_interpolation", 1), ("high_degree_interpolation", 2),
                                                       ("high_degree_interpolation", 3)])
def test_gate_constraints_c_equal_python(kind, param):
    """Gate by gate: honest rows give zeros in both, random rows give the same non-zero values."""
    rng = random.Random(hash((kind, str(param))) & 0xFFFF)
    consts = [rng.randrange(P), rng.randrange(P)]
    pih = [rng.randrange(P) for _ in range(4)]
    width = max(gates_ref.num_wires(kind, param), 1)
    honest = gates_ref.fill_row(kind, param, rng, consts, pih)
    honest = honest + [0] * (width - len(honest))
    rows = [honest, [rng.randrange(P) for _ in range(width)], [rng.randrange(1 << 32) for _ in range(width)]]
    for r, row in enumerate(rows):
        want = gates_ref.constraints(kind, param, consts, row + [0] * 8, pih, gates_ref.Base)
        got = prove_c.gate_constraints(kind, param, consts, row, pih)
        assert got == want, (kind, param, r)
        assert len(got) == gates_ref.num_constraints(kind, param)
        if r == 0:
            assert not any(got)


def test_c_prover_equals_the_golden_proofs():
    """tests/golden/prove_full_2e13.bin and prove_all_gates_2e14.bin were written by the PYTHON prover (gen_prove_golden.py,
    gen_prove_all_gates_golden.py); the C prover reproduces them byte for byte."""
    import ed25519_rows as er

    meta = json.load(open(os.path.join(GOLD, "prove_full_2e13.json")))
    want = open(os.path.join(GOLD, "prove_full_2e13.bin"), "rb").read()
    assert hashlib.sha256(want).hexdigest() == meta["sha256"]
    with accel.c_backend():
        circuit, wires, pis = make_full_circuit(meta["degree_bits"], seed=meta["seed"], arity_bits=tuple(meta["arity_bits"]),
                                                cap_height=meta["cap_height"], num_queries=meta["num_queries"])
    assert prove_c.prove(circuit, wires, pis) == want
    meta = json.load(open(os.path.join(GOLD, "prove_all_gates_2e14.json")))
    want = open(os.path.join(GOLD, "prove_all_gates_2e14.bin"), "rb").read()
    assert hashlib.sha256(want).hexdigest() == meta["sha256"]
    with accel.c_backend():
        circuit, wires, pis = er.make_all_gates_circuit(meta["degree_bits"], seed=meta["seed"], templates=meta["templates"], fri_params=meta["fri_params"])
    c = prove_c.Circuit(circuit)
    assert c.circuit_digest == meta["circuit_digest"]
    assert c.prove(wires, pis) == want
    c.close()
