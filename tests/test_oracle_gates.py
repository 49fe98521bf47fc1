"""Pins oracle/gates_ref.py like the reference pins its gates (gates/gate_testing.rs, per-gate tests):
generated rows satisfy the constraints over the base field and the extension, corrupted rows do not,
constraint counts match Gate::num_constraints, and the base and extension evaluators agree on
base-field inputs."""
import random
import zlib

import pytest

from oracle import gates_ref as g
from oracle import pyref

P = pyref.P
GATES = [("noop", None), ("constant", 2), ("public_input", None), ("arithmetic", 3), ("base_sum", (2, 8)), ("base_sum", (4, 6)),
         ("u32_add_many", (2, 2)), ("u32_add_many", (5, 1)), ("u32_arithmetic", 2), ("u32_subtraction", 2), ("u32_range_check", 2),
         ("comparison", (8, 4)), ("comparison", (32, 16)), ("random_access", (2, 2, 2)), ("random_access", (4, 4, 2)), ("poseidon", None),
         # the eight other gates of upstream plonky2 (round 5), at standard_recursion_config's parameters and small ones
         ("arithmetic_extension", 10), ("mul_extension", 13), ("reducing", 43), ("reducing", 1), ("reducing_extension", 32), ("reducing_extension", 2),
         ("exponentiation", 66), ("exponentiation", 3), ("poseidon_mds", None), ("low_degree_interpolation", 4), ("low_degree_interpolation", 2),
         ("high_degree_interpolation", 2), ("high_degree_interpolation", 3)]


@pytest.mark.parametrize("kind,param", GATES)
def test_generated_rows_satisfy_constraints(kind, param):
    # zlib.crc32, not hash(): str hashes are randomised per process and the rows must be reproducible
    rng = random.Random(zlib.crc32(f"{kind}{param}".encode()))
    for trial in range(3):
        consts = [rng.randrange(P) for _ in range(2)]
        pih = [rng.randrange(P) for _ in range(4)]
        w = g.fill_row(kind, param, rng, consts, pih)
        assert len(w) == g.num_wires(kind, param)
        cb = g.constraints(kind, param, consts, w, pih, g.Base)
        assert len(cb) == g.num_constraints(kind, param)
        assert all(c == 0 for c in cb)
        ce = g.constraints(kind, param, [(c, 0) for c in consts], [(x, 0) for x in w], pih, g.Ext)
        assert all(c == (0, 0) for c in ce)
        if not w:
            continue
        # wires matter: bumping a single wire breaks some constraint (a few wires are legitimately free,
        # e.g. ComparisonGate's equality_dummy of equal chunks)
        sample = rng.sample(range(len(w)), min(len(w), 16))
        broken = 0
        for j in sample:
            bad = list(w)
            bad[j] = (bad[j] + 1 + rng.randrange(P - 1)) % P
            broken += any(c != 0 for c in g.constraints(kind, param, consts, bad, pih, g.Base))
        # legitimately free wires: RandomAccess's unselected list items; ComparisonGate's equality_dummy and
        # intermediate values of equal chunks (all of them when the two inputs are equal)
        floor = {"random_access": 1, "comparison": 0.25 * len(sample)}.get(kind, 0.7 * len(sample))
        assert broken >= floor, (kind, broken, len(sample))


@pytest.mark.parametrize("kind,param", GATES)
def test_base_and_extension_evaluators_agree(kind, param):
    """test_eval_fns (gates/gate_testing.rs:90-150): on random (not satisfying) base inputs the two
    evaluators give the same values; on random extension inputs the result is F_p-linear consistent:
    evaluating at conjugate inputs gives conjugate outputs (X -> -X is an automorphism)."""
    rng = random.Random(7)
    n = max(g.num_wires(kind, param), 1)
    consts = [rng.randrange(P) for _ in range(2)]
    pih = [rng.randrange(P) for _ in range(4)]
    w = [rng.randrange(P) for _ in range(n)]
    cb = g.constraints(kind, param, consts, w, pih, g.Base)
    ce = g.constraints(kind, param, [(c, 0) for c in consts], [(x, 0) for x in w], pih, g.Ext)
    assert ce == [(c, 0) for c in cb]
    we = [(rng.randrange(P), rng.randrange(P)) for _ in range(n)]
    cc = [(rng.randrange(P), rng.randrange(P)) for _ in range(2)]
    conj = lambda v: [(a, (-b) % P) for a, b in v]  # noqa: E731
    assert g.constraints(kind, param, conj(cc), conj(we), pih, g.Ext) == conj(g.constraints(kind, param, cc, we, pih, g.Ext))


def test_poseidon_gate_outputs_are_the_known_answer_permutation():
    from test_oracle_poseidon import TEST_VECTORS

    for inp, exp in TEST_VECTORS:
        # the row for the known input with swap = 0
        seq = iter(list(inp) + [0])
        fake = type("R", (), {"randrange": lambda self, n: next(seq) % n})()
        w = g.fill_row("poseidon", None, fake, [0, 0], [0] * 4)
        assert w[12:24] == [x % P for x in exp]
