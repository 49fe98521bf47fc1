"""The reference's circuit-specific FFI symbol `compute_quotient_polys` (cuda/src/lib.rs:117-143,
cuda/plonky2_gpu.cu:609-783): the ed25519 circuit compiled into the library, buffers in the reference's layout.

Checked on RANDOM leaf data, which is the stronger test for this stage: with random selector columns every one of
the 25 gates' filters is non-zero at every point, so all 231 constraints of all gates contribute everywhere (on a
satisfying witness one gate per row is live and its constraints are zero). The oracle evaluates the same table
with oracle/gates_ref.py + oracle/plonk_ref.py. Bit-exact."""
import os
import random
import sys

import numpy as np
import pytest

from gpu_util import gpu  # noqa: F401

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = 0xFFFFFFFF00000001


def test_committed_gate_program_is_what_the_generator_writes():
    """csrc/ed25519_gate_program.inc is generated data; it must follow the emitters and the circuit table."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_ed25519_program as g

    from plonky2_gpu_amd import ed25519_circuit as ed

    committed = open(os.path.join(ROOT, "plonky2_gpu_amd", "csrc", "ed25519_gate_program.inc")).read()
    assert committed == g.render(), "re-run tools/gen_ed25519_program.py and rebuild"
    assert len(ed.GATES) == len(ed.SELECTOR_INDICES) == 25 and ed.GROUPS[-1][1] == 25
    for row, sel in enumerate(ed.SELECTOR_INDICES):
        assert ed.GROUPS[sel][0] <= row < ed.GROUPS[sel][1]
    assert ed.CONSTANTS_SIGMAS_LEAF_LEN == ed.NUM_CONSTANTS + ed.NUM_ROUTED_WIRES
    assert ed.ZS_PARTIAL_PRODUCTS_LEAF_LEN == ed.NUM_CHALLENGES * (1 + ed.NUM_PARTIAL_PRODUCTS)
    assert ed.NUM_PARTIAL_PRODUCTS == -(-ed.NUM_ROUTED_WIRES // ed.QUOTIENT_DEGREE_FACTOR) - 1


def random_instance(log_len, seed):
    from plonky2_gpu_amd import ed25519_circuit as ed

    rng = np.random.default_rng(seed)
    n_ext = (1 << log_len) << ed.RATE_BITS
    rnd = lambda *shape: rng.integers(0, P, size=shape, dtype=np.uint64)  # noqa: E731
    return dict(log_len=log_len, n_ext=n_ext, wires=rnd(n_ext, ed.NUM_WIRES), cs=rnd(n_ext, ed.CONSTANTS_SIGMAS_LEAF_LEN),
                zs=rnd(n_ext, ed.ZS_PARTIAL_PRODUCTS_LEAF_LEN), k_is=np.array([pow(7, j, P) for j in range(ed.NUM_ROUTED_WIRES)], dtype=np.uint64),
                alphas=rnd(2), betas=rnd(2), gammas=rnd(2))


def run_symbol(gpu, inst):
    import plonky2_gpu_amd as pg

    up = lambda a: pg.DeviceBuffer.from_host(gpu, np.ascontiguousarray(a).reshape(-1))  # noqa: E731
    bufs = {k: up(inst[k]) for k in ("wires", "zs", "cs", "k_is", "alphas", "betas", "gammas")}
    out = pg.reference_compute_quotient_polys(gpu, bufs["wires"], inst["log_len"], bufs["zs"], bufs["cs"], bufs["k_is"], bufs["alphas"],
                                              bufs["betas"], bufs["gammas"])
    return out.download().reshape(2, inst["n_ext"]), bufs


def _lib_call_release():
    from plonky2_gpu_amd import _lib

    _lib.call("gl_reference_quotient_release")


def oracle_quotient(inst, pih):
    from oracle import plonk_ref, prove_ref, pyref
    from plonky2_gpu_amd import ed25519_circuit as ed

    gates = prove_ref.base_gates({"gates": ed.GATES})
    bits = inst["log_len"] + ed.RATE_BITS
    w_l, cs_l, z_l = (inst[k].tolist() for k in ("wires", "cs", "zs"))
    terms = []
    for i in range(inst["n_ext"]):
        row = pyref.reverse_bits(i, bits)
        terms.append(plonk_ref.evaluate_gate_constraints(gates, ed.SELECTOR_INDICES, ed.GROUPS, ed.NUM_GATE_CONSTRAINTS,
                                                         cs_l[row][:ed.NUM_CONSTANTS], w_l[row], list(pih)))
    lst = lambda k: [int(v) for v in inst[k]]  # noqa: E731
    return plonk_ref.compute_quotient_polys(w_l, cs_l, z_l, ed.NUM_CONSTANTS, lst("k_is"), lst("betas"), lst("gammas"), lst("alphas"),
                                            inst["log_len"], ed.RATE_BITS, ed.QUOTIENT_DEGREE_FACTOR, terms)


@pytest.mark.gpu
@pytest.mark.parametrize("log_len", [1, 4])
def test_reference_symbol_equals_the_oracle(gpu, log_len):
    import plonky2_gpu_amd as pg
    from plonky2_gpu_amd import ed25519_circuit as ed

    inst = random_instance(log_len, seed=9000 + log_len)
    got, _ = run_symbol(gpu, inst)
    exp = np.array(oracle_quotient(inst, ed.REFERENCE_PUBLIC_INPUTS_HASH), dtype=np.uint64)
    assert (got == exp).all()
    # another proof of the same circuit: its own public-inputs hash (the PublicInput gate reads it)
    other = [random.Random(5).randrange(P) for _ in range(4)]
    try:
        pg.reference_set_public_inputs_hash(other)
        got2, _ = run_symbol(gpu, inst)
    finally:
        pg.reference_set_public_inputs_hash(None)
    assert (got2 != got).any()
    assert (got2 == np.array(oracle_quotient(inst, other), dtype=np.uint64)).all()
    again, _ = run_symbol(gpu, inst)  # default restored
    assert (again == got).all()
    # the two ways of reading the leaf-major inputs: through the library's column-major staging buffer (default)
    # and in place
    os.environ["PLONKY2_HIP_REFERENCE_IN_PLACE"] = "1"
    try:
        in_place, _ = run_symbol(gpu, inst)
    finally:
        del os.environ["PLONKY2_HIP_REFERENCE_IN_PLACE"]
    assert (in_place == got).all()
    _lib_call_release()
    after_release, _ = run_symbol(gpu, inst)  # the staging buffer comes back on demand
    assert (after_release == got).all()


@pytest.mark.gpu
def test_reference_symbol_equals_the_generic_entry_point(gpu):
    """At 2^13 points: the symbol (compiled-in programs, run-time compiled kernel) against gl_compute_quotient_polys
    given the same circuit as ARGUMENTS and run by the interpreter — two independent routes through the library."""
    import ctypes

    import plonky2_gpu_amd as pg
    from plonky2_gpu_amd import _lib, ed25519_circuit as ed, gate_program as gp

    inst = random_instance(10, seed=9100)
    got, bufs = run_symbol(gpu, inst)
    pool = gp.ImmediatePool()
    prog = pg.GateProgram(gpu, [gp.build_gate(k, p, pool) for k, p in ed.GATES], ed.SELECTOR_INDICES, ed.GROUPS,
                          ed.REFERENCE_PUBLIC_INPUTS_HASH, immediates=pool.values)
    a, b, g = (np.ascontiguousarray(inst[k]) for k in ("alphas", "betas", "gammas"))
    args = _lib.GlQuotientArgs(bufs["wires"].ptr, bufs["cs"].ptr, bufs["zs"].ptr, ed.NUM_WIRES, ed.CONSTANTS_SIGMAS_LEAF_LEN,
                               ed.ZS_PARTIAL_PRODUCTS_LEAF_LEN, bufs["k_is"].ptr, None, b.ctypes.data, g.ctypes.data, a.ctypes.data,
                               ed.NUM_CONSTANTS, ed.NUM_ROUTED_WIRES, 2, ed.NUM_GATE_CONSTRAINTS, 10, ed.RATE_BITS,
                               ed.QUOTIENT_DEGREE_FACTOR, ed.COSET_SHIFT, ctypes.pointer(prog.struct), 0, None, None, None)
    out = pg.DeviceBuffer(gpu, 2 * inst["n_ext"])
    _lib.call("gl_compute_quotient_polys", ctypes.byref(args), out.ptr, gpu.ptr)
    assert (out.download().reshape(2, -1) == got).all()


EDGE_VALUES = [0, 1, 2, 3, 4, 5, (1 << 32) - 2, (1 << 32) - 1, 1 << 32, (1 << 32) + 1, (1 << 63) - 1, 1 << 63, P - 4, P - 3, P - 2, P - 1]
NON_CANONICAL = [P, P + 1, P + 2, P + 3, (1 << 64) - 3, (1 << 64) - 2, (1 << 64) - 1]


def _generic(gpu, inst, bufs, log_len, kernel=None, prog=None, pih=None):
    import ctypes

    import plonky2_gpu_amd as pg
    from plonky2_gpu_amd import _lib, ed25519_circuit as ed

    a, b, g = (np.ascontiguousarray(inst[k]) for k in ("alphas", "betas", "gammas"))
    h = np.array(pih if pih is not None else ed.REFERENCE_PUBLIC_INPUTS_HASH, dtype=np.uint64)
    work = pg.DeviceBuffer(gpu, 2 * inst["n_ext"]) if kernel else None
    args = _lib.GlQuotientArgs(bufs["wires"].ptr, bufs["cs"].ptr, bufs["zs"].ptr, ed.NUM_WIRES, ed.CONSTANTS_SIGMAS_LEAF_LEN,
                               ed.ZS_PARTIAL_PRODUCTS_LEAF_LEN, bufs["k_is"].ptr, None, b.ctypes.data, g.ctypes.data, a.ctypes.data,
                               ed.NUM_CONSTANTS, ed.NUM_ROUTED_WIRES, 2, ed.NUM_GATE_CONSTRAINTS, log_len, ed.RATE_BITS,
                               ed.QUOTIENT_DEGREE_FACTOR, ed.COSET_SHIFT, ctypes.pointer(prog.struct) if prog is not None and not kernel else None, 0,
                               kernel, h.ctypes.data if kernel else None, work.ptr if kernel else None)
    out = pg.DeviceBuffer(gpu, 2 * inst["n_ext"])
    _lib.call("gl_compute_quotient_polys", ctypes.byref(args), out.ptr, gpu.ptr)
    res = out.download().reshape(2, -1)
    if work is not None:
        work.free()
    out.free()
    return res


EDGE_LOG_LEN = 4


def edge_instance(which):
    """the leaves of the test below (also built by its child process, tests/gate_jit_variant_child.py)"""
    inst = random_instance(EDGE_LOG_LEN, seed=9300)
    rng = np.random.default_rng(77)
    edge = np.array(EDGE_VALUES + (NON_CANONICAL if which == "non-canonical" else []), dtype=np.uint64)
    for key in ("wires", "cs"):
        drawn = edge[rng.integers(0, edge.size, size=inst[key].shape)]
        if which == "edges among random":
            inst[key] = np.where(rng.random(inst[key].shape) < 0.02, drawn, inst[key])
        else:
            inst[key] = drawn
    return inst


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["edges", "edges among random", "non-canonical"])
def test_compiled_gates_equal_the_interpreter_where_the_short_forms_take_their_rare_paths(gpu, which):
    """The run-time compiled kernel computes l - 3, t + 2, b - 1 ... with two-instruction forms whose wrap correction sits behind a
    branch, and a base-4 limb's range check as (l (l - 3) + 1)^2 with the constant taken off per gate (csrc/gate_jit.hip, peephole
    pass), once for all the gates of a unit that check the same wire (fused units). On an LDE those wraps need a wire within 3 of
    zero: never. Here the leaves ARE such values — every wire and constant drawn from {0..5, 2^32 +- 1, 2^63, p - 4..p - 1}, the same
    sprinkled into random leaves (so that some lanes of a wave take a correction and others do not), and representatives at and
    above p — and the compiled kernel gives what the interpreter gives, which executes the programs as written; on the canonical
    leaves, also what the oracle's gates give. The generator's two earlier forms (one function per gate, with and without the
    peephole pass: switches of the DIAGNOSTIC build, csrc/knobs.h) are run in a child process each and give the same bytes."""
    import hashlib
    import subprocess

    import plonky2_gpu_amd as pg
    from plonky2_gpu_amd import ed25519_circuit as ed, gate_program as gp

    inst = edge_instance(which)
    got, bufs = run_symbol(gpu, inst)  # the compiled-in table: fused units
    pool = gp.ImmediatePool()
    prog = pg.GateProgram(gpu, [gp.build_gate(k, p, pool) for k, p in ed.GATES], ed.SELECTOR_INDICES, ed.GROUPS,
                          ed.REFERENCE_PUBLIC_INPUTS_HASH, immediates=pool.values)
    interpreted = _generic(gpu, inst, bufs, EDGE_LOG_LEN, prog=prog)
    assert (interpreted == got).all()
    if which != "non-canonical":
        assert (got == np.array(oracle_quotient(inst, ed.REFERENCE_PUBLIC_INPUTS_HASH), dtype=np.uint64)).all()
    prog.compile(ed.NUM_GATE_CONSTRAINTS, 2)
    src = prog.kernel_source()
    assert 0 < src.count("gl::mul_add_small<1>(") < 1000 and "GateSum gate_8()" not in src and "// gate_8" in src
    assert (_generic(gpu, inst, bufs, EDGE_LOG_LEN, kernel=prog.kernel) == got).all()
    # the product library reads no switch: the same source whatever the environment says
    os.environ["PLONKY2_HIP_JIT_FUSE"] = "0"
    try:
        prog.compile(ed.NUM_GATE_CONSTRAINTS, 2)
    finally:
        del os.environ["PLONKY2_HIP_JIT_FUSE"]
    assert prog.kernel_source() == src
    debug_lib = os.path.join(ROOT, "plonky2_gpu_amd", "libplonky2_hip_debug.so")
    assert os.path.exists(debug_lib), "make -C plonky2_gpu_amd/csrc debug (done by __graft_entry__.build())"
    child = os.path.join(ROOT, "tests", "gate_jit_variant_child.py")
    want_sha = hashlib.sha256(np.ascontiguousarray(got).tobytes()).hexdigest()
    for env, marks in (({"PLONKY2_HIP_JIT_FUSE": "0", "PLONKY2_HIP_JIT_PEEPHOLE": "0"}, "0 0 0 1 0"),  # as written, one function per gate (round 4)
                       ({"PLONKY2_HIP_JIT_FUSE": "0"}, "1838 1838 1 1 0")):                      # + the peephole pass
        r = subprocess.run([sys.executable, child, which], env=dict(os.environ, PLONKY2_HIP_LIBRARY=debug_lib, **env), capture_output=True, text=True,
                           timeout=900)
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
        assert "sha256 " + want_sha in r.stdout and "marks " + marks in r.stdout, (env, r.stdout[-500:])


def _poly_at(coeffs, x):
    """sum_j coeffs[j] x^j mod p with vectorised numpy field arithmetic (tools/synth_circuit.py), no oracle arithmetic involved"""
    import synth_circuit as sc

    pw = np.ones(1, dtype=np.uint64)
    for b in range(int(coeffs.size).bit_length() - 1):
        pw = np.concatenate([pw, sc.np_mul(pw, np.uint64(pow(x, 1 << b, P)))])
    terms = sc.np_mul(np.ascontiguousarray(coeffs, dtype=np.uint64), pw)
    while terms.size > 1:
        half = terms.size // 2
        terms = sc.np_add(terms[:half], terms[half:])
    return int(terms[0])


@pytest.mark.gpu
def test_reference_symbol_at_the_size_the_reference_hard_wires(gpu):
    """log_len = 18 is the only size cuda/plonky2_gpu.cu:665-673, 746 supports (values_num_per_poly = 2^18, 2^21 LDE points,
    234 + 88 + 20 leaf elements per point = 5.7 GB of leaves). On random leaves, where all 25 gates are live at every point:
      * the symbol's two quotient polynomials equal gl_compute_quotient_polys given the same circuit as arguments, with the
        run-time compiled kernel AND with the interpreter;
      * at sampled LDE points x the polynomials evaluate to what the oracle computes for that point from the leaves
        (prover.rs:903-991 restated per point: gate constraints, permutation terms, alpha reduction, division by Z_H)."""
    import ctypes
    import sys as _sys

    _sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import plonky2_gpu_amd as pg
    from oracle import plonk_ref, prove_ref, pyref
    from plonky2_gpu_amd import _lib, ed25519_circuit as ed, gate_program as gp

    log_len = 18
    inst = random_instance(log_len, seed=9200)
    n, n_ext, bits = 1 << log_len, inst["n_ext"], log_len + ed.RATE_BITS
    got, bufs = run_symbol(gpu, inst)
    assert got.shape == (2, n_ext)

    # (1) the generic entry point, both device routes
    pool = gp.ImmediatePool()
    prog = pg.GateProgram(gpu, [gp.build_gate(k, p, pool) for k, p in ed.GATES], ed.SELECTOR_INDICES, ed.GROUPS,
                          ed.REFERENCE_PUBLIC_INPUTS_HASH, immediates=pool.values)
    a, b, g = (np.ascontiguousarray(inst[k]) for k in ("alphas", "betas", "gammas"))
    out = pg.DeviceBuffer(gpu, 2 * n_ext)
    pih_host = np.array(ed.REFERENCE_PUBLIC_INPUTS_HASH, dtype=np.uint64)
    for compiled in (False, True):
        work = None
        if compiled:
            prog.compile(ed.NUM_GATE_CONSTRAINTS, 2)   # the ed25519 table's code object comes from build()'s cache
            work = pg.DeviceBuffer(gpu, 2 * n_ext)
        args = _lib.GlQuotientArgs(bufs["wires"].ptr, bufs["cs"].ptr, bufs["zs"].ptr, ed.NUM_WIRES, ed.CONSTANTS_SIGMAS_LEAF_LEN,
                                   ed.ZS_PARTIAL_PRODUCTS_LEAF_LEN, bufs["k_is"].ptr, None, b.ctypes.data, g.ctypes.data, a.ctypes.data,
                                   ed.NUM_CONSTANTS, ed.NUM_ROUTED_WIRES, 2, ed.NUM_GATE_CONSTRAINTS, log_len, ed.RATE_BITS,
                                   ed.QUOTIENT_DEGREE_FACTOR, ed.COSET_SHIFT, None if compiled else ctypes.pointer(prog.struct), 0,
                                   prog.kernel if compiled else None, pih_host.ctypes.data if compiled else None,
                                   work.ptr if compiled else None)
        _lib.call("gl_compute_quotient_polys", ctypes.byref(args), out.ptr, gpu.ptr)
        assert (out.download().reshape(2, -1) == got).all(), "compiled" if compiled else "interpreted"
        if work is not None:
            work.free()

    # (2) the oracle, point by point, on a strided sample of the 2^21 LDE points
    gates = prove_ref.base_gates({"gates": ed.GATES})
    qdb = (ed.QUOTIENT_DEGREE_FACTOR - 1).bit_length()
    assert qdb == ed.RATE_BITS  # the ed25519 configuration: every LDE point is a quotient point
    w = pyref.root_of_unity(bits)
    shift = ed.COSET_SHIFT
    g_pow_n = pow(shift, n, P)
    lst = lambda k: [int(v) for v in inst[k]]  # noqa: E731
    k_is, betas, gammas, alphas = lst("k_is"), lst("betas"), lst("gammas"), lst("alphas")
    pih = list(ed.REFERENCE_PUBLIC_INPUTS_HASH)
    for i in [0, 1, n_ext - 1] + [(977 * 2003 * t + 12345) % n_ext for t in range(9)]:
        row = lambda m: pyref.reverse_bits(m, bits)  # noqa: E731  leaf j holds the point bitrev(j)
        x = shift * pow(w, i, P) % P
        cs = inst["cs"][row(i)].tolist()
        wires = inst["wires"][row(i)].tolist()
        zpp = inst["zs"][row(i)].tolist()
        next_zs = inst["zs"][row((i + (1 << qdb)) % n_ext)].tolist()[:2]
        zh = (g_pow_n * pow(pyref.root_of_unity(qdb), i % (1 << qdb), P) - 1) % P
        l_0_x = zh * plonk_ref.inv(n * (x - 1)) % P
        gate_terms = plonk_ref.evaluate_gate_constraints(gates, ed.SELECTOR_INDICES, ed.GROUPS, ed.NUM_GATE_CONSTRAINTS,
                                                         cs[:ed.NUM_CONSTANTS], wires, pih)
        terms = plonk_ref.vanishing_terms_at(x, l_0_x, wires, cs[ed.NUM_CONSTANTS:ed.NUM_CONSTANTS + ed.NUM_ROUTED_WIRES], zpp[:2], next_zs,
                                             zpp[2:], k_is, betas, gammas, ed.QUOTIENT_DEGREE_FACTOR, gate_terms)
        red = plonk_ref.reduce_with_powers_multi(terms, alphas)
        for c in range(2):
            assert _poly_at(got[c], x) == red[c] * plonk_ref.inv(zh) % P, (i, c)


@pytest.mark.gpu
def test_reference_symbol_rejects_other_shapes(gpu):
    import plonky2_gpu_amd as pg
    from plonky2_gpu_amd import _lib, ed25519_circuit as ed

    n_ext = 8 << ed.RATE_BITS
    big = pg.DeviceBuffer(gpu, n_ext * 256)
    sl = lambda count: _lib.GlDataSlice(big.ptr, count)  # noqa: E731
    good = dict(poly_num=234, n=8, log_len=3, rate_bits=3, zs=sl(n_ext * 20), cs=sl(n_ext * 88), k_is=sl(80), al=sl(2), be=sl(2), ga=sl(2))

    def call(**kw):
        import ctypes

        c = dict(good, **kw)
        ref = ctypes.addressof
        _lib.call("compute_quotient_polys", big.ptr, c["poly_num"], c["n"], c["log_len"], None, None, c["rate_bits"], 0, ref(c["zs"]),
                  ref(c["cs"]), big.ptr, big.ptr, None, None, None, ref(c["k_is"]), ref(c["al"]), ref(c["be"]), ref(c["ga"]), gpu.ptr)

    for bad in (dict(poly_num=135), dict(rate_bits=2), dict(n=16), dict(cs=sl(n_ext * 87)), dict(zs=sl(n_ext * 21)), dict(al=sl(3)),
                dict(k_is=sl(79)), dict(log_len=22, n=1 << 22)):
        with pytest.raises(pg.Plonky2HipError) as e:
            call(**bad)
        assert e.value.code == pg.GL_E_INVALID, bad
