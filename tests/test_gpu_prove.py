"""prove() on the device vs oracle/prove_ref.py: the proof is identical element for element and the
oracle's verifier (which recomputes all challenges from the proof) accepts it."""
import numpy as np
import pytest

from gpu_util import gpu  # noqa: F401
from oracle import prove_ref
from plonk_instance import make_circuit

P = 0xFFFFFFFF00000001

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("degree_bits,two_groups,arity_bits", [(4, False, (2, 1)), (4, True, (1, 2)), (5, True, (3,)), (6, False, (4,))])
def test_prove_equals_oracle_and_verifies(gpu, degree_bits, two_groups, arity_bits):
    import plonky2_gpu_amd as pg

    circuit, wires, pis = make_circuit(degree_bits, seed=3 + degree_bits, two_groups=two_groups, arity_bits=arity_bits)
    cd = pg.CircuitData(gpu, circuit)
    assert cd.constants_sigmas_commitment.merkle_tree.cap.tolist() == circuit["constants_sigmas"]["cap"]
    timing = {}
    proof = pg.prove(gpu, cd, wires, pis, timing)
    assert prove_ref.verify(circuit, proof)
    if degree_bits <= 5:
        exp = prove_ref.prove(circuit, wires, pis)
        for k in ("wires_cap", "plonk_zs_partial_products_cap", "quotient_polys_cap", "openings", "public_inputs"):
            assert proof[k] == exp[k], k
        assert proof["opening_proof"] == exp["opening_proof"]
        # the proof on the wire (util/serialization.rs:674-689) is byte-identical
        from oracle import serialize_ref

        assert pg.serialization.proof_to_bytes(proof) == serialize_ref.proof_bytes(exp)
    assert set(timing) >= {"wires commitment", "quotient polys", "opening proof (FRI)"}


def test_unsatisfied_witness_is_rejected_by_the_verifier(gpu):
    import plonky2_gpu_amd as pg

    circuit, wires, pis = make_circuit(4, seed=5)
    wires = [list(c) for c in wires]
    wires[3] = [(v + 1) % prove_ref.P for v in wires[3]]
    proof = pg.prove(gpu, pg.CircuitData(gpu, circuit), wires, pis)
    with pytest.raises(AssertionError):
        prove_ref.verify(circuit, proof)


@pytest.mark.parametrize("compile_gates", [True, False])
def test_full_gate_list_circuit(gpu, compile_gates):
    """Every gate kind of the ed25519 gate list (BaseSum, U32 add-many / arithmetic / subtraction /
    range-check, Comparison, RandomAccess, Poseidon + the four basic ones) as register programs, run
    by the run-time compiled kernel and by the interpreter: the proof equals the oracle's."""
    import plonky2_gpu_amd as pg
    from oracle import serialize_ref
    from plonk_instance import make_full_circuit

    circuit, wires, pis = make_full_circuit(4, seed=2)
    cd = pg.CircuitData(gpu, circuit, compile_gates=compile_gates)
    proof = pg.prove(gpu, cd, wires, pis)
    assert prove_ref.verify(circuit, proof)
    exp = prove_ref.prove(circuit, wires, pis)
    assert pg.serialization.proof_to_bytes(proof) == serialize_ref.proof_bytes(exp)
    if compile_gates:
        src = cd.gate_program.kernel_source()
        assert "gate_12" in src  # the Poseidon gate is gate 12 of this circuit


@pytest.mark.parametrize("which,compile_gates", [("mini", True), ("mini2", False), ("full", True)])
def test_native_prover_gl_prove(gpu, which, compile_gates):
    """gl_circuit_create + gl_prove (csrc/prove.hip: the host logic in native code): the proof BYTES equal
    the oracle's, for the mini circuits and for the circuit with the whole ed25519 gate list."""
    import plonky2_gpu_amd as pg
    from oracle import serialize_ref
    from plonk_instance import make_full_circuit

    if which == "full":
        circuit, wires, pis = make_full_circuit(4, seed=3)
    else:
        circuit, wires, pis = make_circuit(5 if which == "mini" else 4, seed=9, two_groups=which == "mini2", arity_bits=(2, 1))
    nc = pg.NativeCircuit(gpu, dict(circuit, circuit_digest=None), compile_gates=compile_gates)
    assert nc.circuit_digest == circuit["circuit_digest"]  # derived as circuit_builder.rs:915-927
    assert nc.constants_sigmas_cap == circuit["constants_sigmas"]["cap"]
    timing = {}
    data = nc.prove_bytes(wires, pis, timing)
    exp = prove_ref.prove(circuit, wires, pis)
    assert data == serialize_ref.proof_bytes(exp)
    assert prove_ref.verify(circuit, pg.serialization.proof_from_bytes(data, circuit))
    assert timing["wires commitment"] > 0


@pytest.mark.parametrize("which,degree_bits,arity_bits,cap_height,compile_gates", [
    ("mini", 10, (4, 4), 4, True), ("mini2", 11, (3, 2, 1), 3, False), ("mini", 12, (4, 4, 4), 2, True), ("full", 10, (4, 4), 4, True)])
def test_proof_bytes_equal_the_oracle_at_2e10_to_2e12_rows(gpu, which, degree_bits, arity_bits, cap_height, compile_gates):
    """Byte equality with the CPU restatement of prove() (plonk/prover.rs:41-237) beyond toy sizes: 2^10..2^12 rows, FRI
    arities [4,4] / [3,2,1] / [4,4,4], 28 query rounds, two-pass NTT sizes on the device (LDE 2^13..2^15), the 13-gate
    circuit with every gate kind of the ed25519 list at 2^10 rows. The oracle's Poseidon / Merkle / NTT come from the C
    restatement (oracle/accel.py; same proofs as the pure model, tests/test_oracle_prove.py::test_c_backend_gives_the_same_proof).
    Both provers: the native gl_prove and the Python mirror."""
    import plonky2_gpu_amd as pg
    from oracle import accel, serialize_ref
    from plonk_instance import make_full_circuit

    with accel.c_backend():
        if which == "full":
            circuit, wires, pis = make_full_circuit(degree_bits, seed=5, arity_bits=arity_bits, cap_height=cap_height, num_queries=28)
        else:
            circuit, wires, pis = make_circuit(degree_bits, seed=40 + degree_bits, two_groups=which == "mini2", arity_bits=arity_bits,
                                               cap_height=cap_height, num_queries=28, pow_bits=8)
        exp = prove_ref.prove(circuit, wires, pis)
        assert prove_ref.verify(circuit, exp)
    exp_bytes = serialize_ref.proof_bytes(exp)
    nc = pg.NativeCircuit(gpu, dict(circuit, circuit_digest=None), compile_gates=compile_gates)
    assert nc.circuit_digest == circuit["circuit_digest"]
    data = nc.prove_bytes(wires, pis)
    assert len(data) == len(exp_bytes)
    assert data == exp_bytes
    nc.close()
    if which != "full":
        proof = pg.prove(gpu, pg.CircuitData(gpu, circuit), wires, pis)
        assert pg.serialization.proof_to_bytes(proof) == exp_bytes


def test_proof_bytes_at_2e13_rows_equal_the_fixture(gpu):
    """2^13 rows of the 13-gate circuit (135 wires): the first size at which the commitments inside gl_prove take the pipelined
    branch (48+ columns, 2^16 leaves) and the LDE has 2^16 points per column. The oracle prover needs minutes for it, so its
    proof is a fixture (tests/golden/prove_full_2e13.bin, written by tests/golden/gen_prove_golden.py); circuit and witness are
    rebuilt here from the same seed and must give the same circuit digest and the same proof, byte for byte."""
    import hashlib
    import json
    import os

    import plonky2_gpu_amd as pg
    from oracle import accel
    from plonk_instance import make_full_circuit

    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    meta = json.load(open(os.path.join(gold, "prove_full_2e13.json")))
    want = open(os.path.join(gold, "prove_full_2e13.bin"), "rb").read()
    assert hashlib.sha256(want).hexdigest() == meta["sha256"] and len(want) == meta["bytes"]
    with accel.c_backend():
        circuit, wires, pis = make_full_circuit(meta["degree_bits"], seed=meta["seed"], arity_bits=tuple(meta["arity_bits"]),
                                                cap_height=meta["cap_height"], num_queries=meta["num_queries"])
    assert [int(v) for v in circuit["circuit_digest"]] == meta["circuit_digest"]
    for compile_gates in (True, False):
        nc = pg.NativeCircuit(gpu, dict(circuit, circuit_digest=None), compile_gates=compile_gates)
        assert [int(v) for v in nc.circuit_digest] == meta["circuit_digest"]
        data = nc.prove_bytes(wires, pis)
        nc.close()
        assert len(data) == len(want)
        assert data == want, compile_gates


@pytest.mark.parametrize("qdf,two_groups", [(5, False), (6, False), (4, True)])
def test_quotient_degree_factor_that_is_not_a_power_of_two(gpu, qdf, two_groups):
    """The trimmed-and-copied chunk path (prover.rs:153-166) of both provers — the Python mirror and the
    native gl_prove — against the oracle's proof bytes."""
    import plonky2_gpu_amd as pg
    from oracle import serialize_ref

    circuit, wires, pis = make_circuit(4, seed=20 + qdf, two_groups=two_groups, quotient_degree_factor=qdf)
    exp = serialize_ref.proof_bytes(prove_ref.prove(circuit, wires, pis))
    proof = pg.prove(gpu, pg.CircuitData(gpu, circuit), wires, pis)
    assert pg.serialization.proof_to_bytes(proof) == exp
    nc = pg.NativeCircuit(gpu, dict(circuit, circuit_digest=None))
    assert nc.prove_bytes(wires, pis) == exp


def test_unsatisfied_witness_fails_loudly_when_the_quotient_is_trimmed(gpu):
    """prover.rs:161-165 `expect("Quotient has failed, the vanishing polynomial is not divisible by Z_H")`:
    with quotient_degree_factor = 5 the 8n-coefficient quotient of a broken witness has a non-zero tail."""
    import plonky2_gpu_amd as pg
    from plonky2_gpu_amd._lib import Plonky2HipError

    circuit, wires, pis = make_circuit(4, seed=25, quotient_degree_factor=5)
    bad = [list(c) for c in wires]
    bad[3] = [(v + 1) % prove_ref.P for v in bad[3]]
    with pytest.raises(AssertionError, match="Quotient has failed"):
        prove_ref.prove(circuit, bad, pis)
    with pytest.raises(ValueError, match="Quotient has failed"):
        pg.prove(gpu, pg.CircuitData(gpu, circuit), bad, pis)
    with pytest.raises(Plonky2HipError, match="Quotient has failed"):
        pg.NativeCircuit(gpu, dict(circuit, circuit_digest=None)).prove_bytes(bad, pis)


@pytest.mark.parametrize("num_challenges", [1, 3])
def test_other_numbers_of_challenges(gpu, num_challenges):
    """num_challenges is a loop bound in the partial-product / quotient / FRI code and a compile-time
    constant of the run-time compiled gate kernels; both provers against the oracle's proof bytes."""
    import plonky2_gpu_amd as pg
    from oracle import serialize_ref

    circuit, wires, pis = make_circuit(4, seed=40 + num_challenges, num_challenges=num_challenges)
    exp = serialize_ref.proof_bytes(prove_ref.prove(circuit, wires, pis))
    assert pg.serialization.proof_to_bytes(pg.prove(gpu, pg.CircuitData(gpu, circuit), wires, pis)) == exp
    assert pg.NativeCircuit(gpu, dict(circuit, circuit_digest=None)).prove_bytes(wires, pis) == exp


@pytest.mark.parametrize("degree_bits,arity_bits", [(3, ()), (4, (1, 1, 1)), (6, (4,))])
def test_native_prover_fri_shapes(gpu, degree_bits, arity_bits):
    """gl_prove with no commit-phase layer at all (the final polynomial is the whole combined polynomial),
    with three binary layers (4-element leaves: the `<= 4 elements are not hashed` rule of hash_or_noop),
    and with one 16-ary layer."""
    import plonky2_gpu_amd as pg
    from oracle import serialize_ref

    circuit, wires, pis = make_circuit(degree_bits, seed=50 + degree_bits, arity_bits=arity_bits)
    exp = prove_ref.prove(circuit, wires, pis)
    assert len(exp["opening_proof"]["final_poly"]) == 1 << (degree_bits - sum(arity_bits))
    data = pg.NativeCircuit(gpu, dict(circuit, circuit_digest=None)).prove_bytes(wires, pis)
    assert data == serialize_ref.proof_bytes(exp)


def test_a_2e14_row_proof_is_accepted_by_the_oracle_verifier(gpu):
    """Size-independent property at a size the Python prover cannot reach: a 2^14-row, 135-wire circuit
    (standard_recursion_config's shape: 80 routed wires, rate 8, cap height 4, arities [4, 4, 4], 28 queries,
    16 proof-of-work bits; witness from tools/synth_circuit.py) proven by gl_prove; the oracle's verifier
    recomputes every challenge from the proof bytes and checks vanishing(zeta) = Z_H(zeta) t(zeta), all
    Merkle paths, the folding consistency and the final polynomial. A corrupted byte is rejected."""
    import os
    import sys

    import plonky2_gpu_amd as pg
    from plonky2_gpu_amd.challenger import hash_no_pad

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
    import synth_circuit

    circuit, wires, pis = synth_circuit.make(14, num_wires=135, num_routed=80, num_constants=8, seed=5)
    assert circuit["fri_params"]["reduction_arity_bits"] == [4, 4, 4]
    synth_circuit.set_public_input_row(wires, hash_no_pad(gpu, pis))
    nc = pg.NativeCircuit(gpu, dict(circuit, circuit_digest=None))
    data = nc.prove_bytes(wires, pis)
    vc = dict(circuit, circuit_digest=nc.circuit_digest, constants_sigmas=dict(cap=nc.constants_sigmas_cap))
    assert prove_ref.verify(vc, pg.serialization.proof_from_bytes(data, circuit))
    bad = bytearray(data)
    bad[len(bad) // 3] ^= 1
    with pytest.raises((AssertionError, ValueError)):
        prove_ref.verify(vc, pg.serialization.proof_from_bytes(bytes(bad), circuit))


def test_synthetic_circuit_with_the_ed25519_gate_table_proof_bytes(gpu):
    """The kind of circuit bench.py proves (tools/synth_circuit.py gate_table="ed25519": all 25 gates of the ed25519
    table declared, four kinds instantiated) at 2^4 rows: gl_prove's bytes equal the oracle prover's, with the gates
    run-time compiled and interpreted."""
    import os
    import sys

    import plonky2_gpu_amd as pg
    from oracle import pyref, serialize_ref

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
    import synth_circuit

    fp = dict(rate_bits=3, cap_height=1, reduction_arity_bits=[2], proof_of_work_bits=2, num_query_rounds=2)
    circuit, wires, pis = synth_circuit.make(4, num_wires=234, num_routed=80, num_constants=8, seed=6, fri_params=fp, gate_table="ed25519")
    synth_circuit.set_public_input_row(wires, pyref.hash_no_pad(pis))
    oc = dict(circuit, constants=[[int(v) for v in c] for c in circuit["constants"]], sigmas=[[int(v) for v in c] for c in circuit["sigmas"]])
    oc["constants_sigmas"] = prove_ref.commit_from_values(oc["constants"] + oc["sigmas"], 3, 1)
    oc["circuit_digest"] = prove_ref.circuit_digest(oc["constants_sigmas"]["cap"], 4)
    exp = serialize_ref.proof_bytes(prove_ref.prove(oc, [[int(v) for v in c] for c in wires], pis))
    for compile_gates in (True, False):
        nc = pg.NativeCircuit(gpu, dict(circuit, circuit_digest=None), compile_gates=compile_gates)
        assert nc.circuit_digest == oc["circuit_digest"]
        assert nc.prove_bytes(wires, pis) == exp, compile_gates
        nc.close()


@pytest.mark.parametrize("degree_bits,arity_bits,cap_height,num_queries,compile_gates", [(5, (2, 1), 1, 2, True), (5, (2, 1), 1, 2, False), (11, (4, 4), 4, 28, True)])
def test_recursion_shaped_circuit_with_the_upstream_gate_kinds_proof_bytes(gpu, degree_bits, arity_bits, cap_height, num_queries, compile_gates):
    """f4 widened (round 5): a circuit of standard_recursion_config's shape — 135 wires, 80 routed, 15 gates in 4 selector groups — whose
    rows use the eight gate kinds of upstream plonky2 beyond the ed25519 list (ArithmeticExtension{10}, MulExtension{13}, Reducing{43},
    ReducingExtension{32}, Exponentiation{66}, PoseidonMds, LowDegreeInterpolation{4}, HighDegreeInterpolation{2}: gates/*.rs) next to
    the basic ones, each row honestly generated: gl_prove's bytes (programs from the emitters, run-time compiled and interpreted) equal
    the C restatement of prove(), and the oracle's verifier — extension gates over the D = 2 extension algebra — accepts them."""
    import plonky2_gpu_amd as pg
    from oracle import accel, prove_c
    from plonk_instance import make_recursion_circuit

    with accel.c_backend():
        circuit, wires, pis = make_recursion_circuit(degree_bits, seed=3, arity_bits=arity_bits, cap_height=cap_height, num_queries=num_queries,
                                                     pow_bits=3 if degree_bits < 8 else 10)
    want = prove_c.prove(circuit, wires, pis)
    nc = pg.NativeCircuit(gpu, dict(circuit, circuit_digest=None), compile_gates=compile_gates)
    assert nc.circuit_digest == circuit["circuit_digest"]
    data = nc.prove_bytes(wires, pis)
    nc.close()
    assert data == want
    with accel.c_backend():
        assert prove_ref.verify(circuit, pg.serialization.proof_from_bytes(data, circuit))


def test_a_circuit_description_of_another_header_version_is_refused(gpu):
    """GlCircuitDesc embeds GlFriParams by value and has grown before: its first field is sizeof(GlCircuitDesc) of the caller's header, and
    gl_circuit_create refuses any other value (a caller compiled against the 0.3 layout passes its degree_bits there)."""
    import ctypes

    import plonky2_gpu_amd as pg
    from plonky2_gpu_amd import _lib

    circuit, wires, pis = make_circuit(4, seed=3)
    nc = pg.NativeCircuit(gpu, dict(circuit, circuit_digest=None))  # a good description is accepted
    nc.close()
    desc = _lib.GlCircuitDesc()
    desc.struct_size = 4  # what a stale caller's degree_bits would look like
    h = ctypes.c_void_p()
    with pytest.raises(pg.Plonky2HipError, match="null pointer|struct_size"):
        _lib.call("gl_circuit_create", ctypes.byref(desc), ctypes.byref(h), gpu.ptr)
    k = np.zeros(16, dtype=np.uint64)
    desc.h_k_is = desc.h_constants = desc.h_sigmas = k.ctypes.data
    with pytest.raises(pg.Plonky2HipError, match="struct_size"):
        _lib.call("gl_circuit_create", ctypes.byref(desc), ctypes.byref(h), gpu.ptr)
    assert pg.load().gl_version().startswith(b"plonky2_hip 0.6")


def test_working_buffers_are_recycled_and_can_be_trimmed(gpu):
    """gl_prove keeps one proof's working buffers attached to the circuit; proofs are deterministic across
    the recycled buffers (nothing depends on stale contents) and across gl_circuit_trim."""
    import plonky2_gpu_amd as pg

    circuit, wires, pis = make_circuit(5, seed=61, two_groups=True)
    nc = pg.NativeCircuit(gpu, dict(circuit, circuit_digest=None))
    first = nc.prove_bytes(wires, pis)
    assert nc.prove_bytes(wires, pis) == first  # second proof runs entirely on recycled buffers
    # different public inputs change the transcript from the start: the buffers really are rewritten
    assert nc.prove_bytes(wires, [(x + 1) % prove_ref.P for x in pis]) != first
    nc.trim()
    assert nc.prove_bytes(wires, pis) == first


@pytest.mark.parametrize("degree_bits,proof_bytes", [(18, 204544), (20, 219072)])
def test_full_size_proof_bytes_equal_the_c_oracle(gpu, degree_bits, proof_bytes):
    """The shape bench.py times (BASELINE.json configs[3]: n = 2^18, 234 wires / 80 routed, 88 preprocessed
    polynomials, rate 8, cap height 4, FRI arities [4, 4, 4, 4], 28 queries, 16 proof-of-work bits; the whole 25-gate
    ed25519 table evaluated at every point) proven by gl_prove and compared BYTE FOR BYTE with the C restatement of
    prove() (oracle/prove_oracle.c: plonk/prover.rs:40-233 on the box's CPU cores — the same function that is bench.py's
    prove() CPU baseline), then checked by the oracle's verifier, which recomputes every challenge from the bytes.
    bench.py itself may not use the oracle for this (only its cpu_baseline leg may), so the validity of what it times is
    established here; the witness comes from the same generator with the same seed as the bench's rank 0. The second case,
    2^20 rows (LDE 2^23: three-pass transforms, 15.7 GB of wire LDE), is north_star's 2^20-row trace as a whole proof."""
    import os
    import sys
    import time

    import numpy as np

    import plonky2_gpu_amd as pg
    from oracle import accel, prove_c
    from plonky2_gpu_amd.challenger import hash_no_pad

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
    import synth_circuit

    circuit, wires, pis = synth_circuit.make(degree_bits, num_wires=234, num_routed=80, num_constants=8, seed=1, gate_table="ed25519")
    assert circuit["fri_params"]["reduction_arity_bits"] == [4] * ((degree_bits - 2) // 4)
    assert len(circuit["gates"]) == 25 and circuit["num_gate_constraints"] == 231  # the whole ed25519 gate table is declared
    synth_circuit.set_public_input_row(wires, hash_no_pad(gpu, pis))
    nc = pg.NativeCircuit(gpu, dict(circuit, circuit_digest=None))
    d_wires = pg.DeviceBuffer.from_host(gpu, np.ascontiguousarray(wires))
    data = nc.prove_bytes(d_wires, pis)
    assert len(data) == proof_bytes
    digest, cs_cap = nc.circuit_digest, nc.constants_sigmas_cap
    nc.close()
    d_wires.free()
    t0 = time.perf_counter()
    oc = prove_c.Circuit(circuit)
    assert oc.circuit_digest == [int(v) for v in digest]
    tr = {}
    want = oc.prove(wires, pis, trace=tr)
    oc.close()
    print("C oracle prove() at 2^%d rows on %d threads: %.1f s; stages %s" % (degree_bits, oc.threads, time.perf_counter() - t0,
                                                                            {k: round(v, 1) for k, v in tr["stage_seconds"].items()}))
    assert len(data) == len(want)
    assert data == want, "first differing 8-byte word: %d" % next(i // 8 for i in range(0, len(want), 8) if data[i:i + 8] != want[i:i + 8])
    vc = dict(circuit, circuit_digest=digest, constants_sigmas=dict(cap=cs_cap))
    with accel.c_backend():
        assert prove_ref.verify(vc, pg.serialization.proof_from_bytes(data, circuit))


@pytest.mark.parametrize("compile_gates", [True, False])
def test_all_25_ed25519_gates_with_honest_rows_proof_bytes(gpu, compile_gates):
    """configs[3]'s gate table doing real work: the 25 gates of the ed25519 circuit with their real parameters, every kind
    instantiated by honestly generated rows (tests/ed25519_rows.py), 234 wires, copy constraints — 2^8 rows so that the
    CPU restatement of prove() can be run beside it: gl_prove's bytes equal the oracle's, compiled and interpreted gates."""
    import plonky2_gpu_amd as pg
    from oracle import accel, serialize_ref
    import ed25519_rows as er

    fp = dict(rate_bits=3, cap_height=2, reduction_arity_bits=[3, 2], proof_of_work_bits=4, num_query_rounds=6)
    with accel.c_backend():
        circuit, wires, pis = er.make_all_gates_circuit(8, seed=4, templates=3, fri_params=fp)
        oc, ow = er.as_oracle_circuit(circuit, wires, prove_ref)
        exp = prove_ref.prove(oc, ow, pis)
        assert prove_ref.verify(oc, exp)
    nc = pg.NativeCircuit(gpu, dict(circuit, circuit_digest=None), compile_gates=compile_gates)
    assert nc.circuit_digest == oc["circuit_digest"]
    assert nc.prove_bytes(wires, pis) == serialize_ref.proof_bytes(exp)
    nc.close()


def test_all_25_gates_proof_bytes_at_2e14_rows_equal_the_fixture(gpu):
    """The all-25-gates circuit (tests/ed25519_rows.py) at 2^14 rows — 655 honest rows per gate kind, 234 wires, LDE 2^17,
    pipelined commits, two-pass transforms — byte for byte against the oracle's proof kept as a fixture
    (tests/golden/prove_all_gates_2e14.bin, written by tests/golden/gen_prove_all_gates_golden.py in a quarter of an hour);
    compiled and interpreted gates. Until round 3 this circuit was byte-checked at 2^8 rows only."""
    import hashlib
    import json
    import os

    import ed25519_rows as er
    import plonky2_gpu_amd as pg
    from oracle import accel

    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    meta = json.load(open(os.path.join(gold, "prove_all_gates_2e14.json")))
    want = open(os.path.join(gold, "prove_all_gates_2e14.bin"), "rb").read()
    assert hashlib.sha256(want).hexdigest() == meta["sha256"] and len(want) == meta["bytes"]
    with accel.c_backend():
        circuit, wires, pis = er.make_all_gates_circuit(meta["degree_bits"], seed=meta["seed"], templates=meta["templates"],
                                                        fri_params=meta["fri_params"])
    for compile_gates in (True, False):
        nc = pg.NativeCircuit(gpu, dict(circuit, circuit_digest=None), compile_gates=compile_gates)
        assert [int(v) for v in nc.circuit_digest] == meta["circuit_digest"]
        data = nc.prove_bytes(wires, pis)
        nc.close()
        assert data == want, compile_gates


def test_full_size_proof_with_all_25_gate_kinds_in_use_bytes_equal_the_c_oracle(gpu):
    """BASELINE.json configs[3] at its full shape (2^18 rows, 234 wires / 80 routed, 88 preprocessed polynomials, rate 8,
    cap height 4, arities [4,4,4,4], 28 queries, 16 PoW bits) with EVERY one of the 25 gate kinds constraining rows
    (10 485 rows each, honestly generated, tied by copy constraints): proven by gl_prove, BYTE-EQUAL to the C restatement of
    prove() (oracle/prove_oracle.c), accepted by the oracle's verifier, which re-derives every challenge and evaluates all 25
    gates' constraints at zeta over F_p^2 on its own."""
    import numpy as np

    import plonky2_gpu_amd as pg
    from plonky2_gpu_amd.challenger import hash_no_pad
    from oracle import accel
    import ed25519_rows as er

    with accel.c_backend():
        circuit, wires, pis = er.make_all_gates_circuit(18, seed=1, templates=4, pih_of=lambda x: hash_no_pad(gpu, x))
    assert circuit["fri_params"]["reduction_arity_bits"] == [4, 4, 4, 4]
    nc = pg.NativeCircuit(gpu, dict(circuit, circuit_digest=None))
    d_wires = pg.DeviceBuffer.from_host(gpu, np.ascontiguousarray(wires))
    data = nc.prove_bytes(d_wires, pis)
    assert len(data) == 204544
    from oracle import prove_c

    assert data == prove_c.prove(circuit, wires, pis), "gl_prove's bytes differ from the C restatement of prove() at 2^18 rows with all 25 gate kinds in use"
    vc = dict(circuit, circuit_digest=nc.circuit_digest, constants_sigmas=dict(cap=nc.constants_sigmas_cap))
    with accel.c_backend():
        assert prove_ref.verify(vc, pg.serialization.proof_from_bytes(data, circuit))
    # the same circuit with one limb of one U32RangeCheckGate{8} row off by one: the quotient stops being a polynomial
    bad = wires.copy()
    row = next(r for r in range(1 << 18) if (r + 2) % 25 == 21)
    bad[10, row] = (int(bad[10, row]) + 1) % prove_ref.P
    from plonky2_gpu_amd._lib import Plonky2HipError
    try:
        data_bad = nc.prove_bytes(pg.DeviceBuffer.from_host(gpu, np.ascontiguousarray(bad)), pis)
    except Plonky2HipError:
        data_bad = None  # "Quotient has failed" is also a legitimate way to notice
    if data_bad is not None:
        with accel.c_backend(), pytest.raises((AssertionError, ValueError)):
            prove_ref.verify(vc, pg.serialization.proof_from_bytes(data_bad, circuit))
    nc.close()


def test_witness_upload_overlapping_a_proof(gpu):
    """gl_memcpy_h2d_async: the next witness travels on the context's second stream while gl_prove runs on the
    first; after ctx.synchronize() it is complete and proves to the same bytes as a witness uploaded up front, and
    the proof that ran meanwhile is unaffected."""
    import numpy as np

    import plonky2_gpu_amd as pg

    circuit, wires, pis = make_circuit(7, seed=71, two_groups=True)
    other, wires2, pis2 = make_circuit(7, seed=72, two_groups=True)
    nc = pg.NativeCircuit(gpu, dict(circuit, circuit_digest=None))
    flat = np.ascontiguousarray(np.array(wires, dtype=np.uint64).reshape(-1))
    expect = nc.prove_bytes(wires, pis)
    staging = pg.PinnedArray(flat.size)
    staging.array[:] = flat
    d_a = pg.DeviceBuffer.from_host(gpu, np.array(wires2, dtype=np.uint64).reshape(-1))  # some other witness of the same shape
    d_b = pg.DeviceBuffer(gpu, flat.size)
    nc2 = pg.NativeCircuit(gpu, dict(other, circuit_digest=None))
    meanwhile_expect = nc2.prove_bytes(d_a, pis2)
    d_b.upload_async(staging)
    meanwhile = nc2.prove_bytes(d_a, pis2)
    gpu.synchronize()
    assert meanwhile == meanwhile_expect
    assert (d_b.download() == flat).all()
    assert nc.prove_bytes(d_b, pis) == expect
    staging.free()


def test_two_host_threads_with_their_own_contexts_on_one_device(gpu):
    """Two threads, each with its own context, circuit and data, hammering natural-order transforms (which stage through the
    workspace), commits with the leaf-major copy (the event pair) and whole proofs at the same time must each get exactly the
    results they get alone. Up to round 5 the workspace and the event pair existed once per device and the entry points took
    turns (a per-device lock); since round 6 they belong to the context (capi.hip CtxState) and the two threads' calls really
    run at the same time — tests/test_gpu_contexts.py holds that they do."""
    import threading

    import numpy as np

    import plonky2_gpu_amd as pg

    def work(ctx, seed, rounds, out):
        rng = np.random.default_rng(seed)
        x = rng.integers(0, prove_ref.P, size=(3, 1 << 13), dtype=np.uint64)
        vals = rng.integers(0, prove_ref.P, size=(5, 1 << 9), dtype=np.uint64)
        circuit, wires, pis = make_circuit(6, seed=80 + seed, two_groups=bool(seed & 1))
        nc = pg.NativeCircuit(ctx, dict(circuit, circuit_digest=None))
        for _ in range(rounds):
            f = pg.fft_with_options(ctx, x)
            back = pg.ifft_with_options(ctx, f)
            b = pg.PolynomialBatch.from_values(ctx, vals, 3, False, 2, leaf_major=True)
            out.append((f.tobytes(), back.tobytes(), b.merkle_tree.cap.tobytes(), b.merkle_tree.digests.tobytes(), nc.prove_bytes(wires, pis)))
        nc.close()

    other = pg.Context(0)
    try:
        alone = [[], []]
        work(gpu, 1, 1, alone[0])
        work(other, 2, 1, alone[1])
        together = [[], []]
        threads = [threading.Thread(target=work, args=(c, s, 6, o)) for c, s, o in ((gpu, 1, together[0]), (other, 2, together[1]))]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        for who in (0, 1):
            assert len(together[who]) == 6
            for got in together[who]:
                assert got == alone[who][0], who
    finally:
        other.close()


def test_device_memory_does_not_grow_over_proofs_circuits_and_commits(gpu):
    """A prover is a long-lived process: free device memory must be the same after many proofs on one circuit
    (the per-circuit buffer pool recycles), after creating and destroying circuits, and after commits of changing
    shapes (coset-table cache churn) as it was after the first few."""
    import ctypes

    import numpy as np

    import plonky2_gpu_amd as pg

    hip = ctypes.CDLL("libamdhip64.so")

    def free_bytes():
        gpu.synchronize()
        f, t = ctypes.c_size_t(), ctypes.c_size_t()
        assert hip.hipMemGetInfo(ctypes.byref(f), ctypes.byref(t)) == 0
        return f.value

    circuit, wires, pis = make_circuit(6, seed=90, two_groups=True)
    base = None
    for i in range(24):
        nc = pg.NativeCircuit(gpu, dict(circuit, circuit_digest=None))
        nc.prove_bytes(wires, pis)
        nc.close()
        if i == 3:
            base = free_bytes()
    # ">=": creating and destroying a circuit also loads and unloads its gate-kernel modules (eight code objects since round 3), and the
    # HIP runtime returns pooled code-object memory when it likes: more free memory than at the reference point is not a leak
    assert free_bytes() >= base, "circuits leak device memory"
    nc = pg.NativeCircuit(gpu, dict(circuit, circuit_digest=None))
    first = nc.prove_bytes(wires, pis)
    base = free_bytes()
    for _ in range(100):
        assert nc.prove_bytes(wires, pis) == first
    assert free_bytes() == base, "proofs leak device memory"
    nc.close()
    # the error path gives its buffers back too, and leaves the circuit usable
    c5, w5, p5 = make_circuit(6, seed=91, two_groups=True, quotient_degree_factor=5)
    nc = pg.NativeCircuit(gpu, dict(c5, circuit_digest=None))
    good = nc.prove_bytes(w5, p5)
    base = free_bytes()
    bad = [list(c) for c in w5]
    bad[3][7] = (bad[3][7] + 1) % prove_ref.P
    for _ in range(5):
        with pytest.raises(pg.Plonky2HipError, match="not divisible by Z_H"):
            nc.prove_bytes(bad, p5)
        assert nc.prove_bytes(w5, p5) == good
    assert free_bytes() == base, "failing proofs leak device memory"
    nc.close()
    rng = np.random.default_rng(4)
    base = None
    for i in range(80):
        log_n, cols = int(rng.integers(3, 11)), int(rng.integers(1, 12))
        pg.PolynomialBatch.from_values(gpu, rng.integers(0, prove_ref.P, size=(cols, 1 << log_n), dtype=np.uint64), int(rng.integers(1, 4)), False, 1)
        if i == 40:
            base = free_bytes()
    import gc

    gc.collect()
    assert free_bytes() >= base, "commits leak device memory"


@pytest.mark.parametrize("which,degree_bits,compile_gates", [("mini", 5, True), ("mini2", 4, False), ("full", 4, True), ("mini", 10, True)])
def test_blinded_proof_bytes_equal_the_oracle(gpu, which, degree_bits, compile_gates):
    """gl_prove_zk: a circuit with fri_params.hiding (CircuitConfig::zero_knowledge, plonk/circuit_data.rs:74) — the wires, Zs /
    partial products and quotient commitments carry SALT_SIZE = 4 caller-supplied random elements per leaf (plonk/prover.rs:84, 125,
    174; fri/oracle.rs:985-1002) — gives, with the same salts, the oracle's proof byte for byte; the oracle's verifier accepts it;
    gl_prove refuses a hiding circuit and gl_prove_zk a plain one."""
    import plonky2_gpu_amd as pg
    from oracle import serialize_ref
    from plonk_instance import make_full_circuit

    if which == "full":
        circuit, wires, pis = make_full_circuit(degree_bits, seed=5)
    else:
        circuit, wires, pis = make_circuit(degree_bits, seed=14, two_groups=which == "mini2", arity_bits=(2, 1) if degree_bits < 8 else (4, 4))
    plain = circuit
    circuit = dict(circuit, fri_params=dict(circuit["fri_params"], hiding=True))
    n_ext = 1 << (circuit["degree_bits"] + circuit["fri_params"]["rate_bits"])
    salts = np.random.default_rng(99 + degree_bits).integers(0, P, size=(3, 4, n_ext), dtype=np.uint64)
    nc = pg.NativeCircuit(gpu, dict(circuit, circuit_digest=None), compile_gates=compile_gates)
    data = nc.prove_bytes(wires, pis, salts=salts)
    if degree_bits <= 5:  # the pure-Python model; above it the C-accelerated one (same proofs, tests/test_oracle_prove.py)
        exp = prove_ref.prove(circuit, wires, pis, salts=salts.tolist())
    else:
        from oracle import accel

        with accel.c_backend():
            exp = prove_ref.prove(circuit, wires, pis, salts=salts.tolist())
    assert data == serialize_ref.proof_bytes(exp)
    parsed = pg.serialization.proof_from_bytes(data, circuit)
    assert prove_ref.verify(circuit, parsed)
    assert len(parsed["opening_proof"]["query_round_proofs"][0]["initial_trees_proof"][1][0]) == circuit["num_wires"] + 4
    # other salts, other commitments; the flag and the entry point go together
    other = nc.prove_bytes(wires, pis, salts=(salts + np.uint64(1)) % np.uint64(P))
    assert other[:32] != data[:32] and prove_ref.verify(circuit, pg.serialization.proof_from_bytes(other, circuit))
    with pytest.raises(pg.Plonky2HipError, match="hiding"):
        nc.prove_bytes(wires, pis)
    # salts given as RAW 64-bit words (a caller filling d_salts from a byte stream): about one word in 2^32 is >= p, here a few are
    # forced to be. They are reduced on their way into the commitment: the proof's bytes stay canonical (util/serialization.rs:492-497
    # writes to_canonical_u64) and equal the proof for the reduced salts.
    raw, small = salts.copy(), salts.copy()
    small[:, :, :5] = salts[:, :, :5] % np.uint64(0xFFFFFFFF)
    raw[:, :, :5] = small[:, :, :5] + np.uint64(P)
    assert (raw >= np.uint64(P)).sum() >= 60
    assert nc.prove_bytes(wires, pis, salts=raw) == nc.prove_bytes(wires, pis, salts=small)
    nc.close()
    nc2 = pg.NativeCircuit(gpu, dict(plain, circuit_digest=None), compile_gates=compile_gates)
    with pytest.raises(pg.Plonky2HipError, match="not hiding"):
        nc2.prove_bytes(wires, pis, salts=salts)
    nc2.close()
