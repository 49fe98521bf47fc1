"""prove_c.py — ctypes binding of oracle/prove_oracle.c, the C restatement of prove() above the commit
(TEST INFRASTRUCTURE ONLY; only tests/, smoke() and bench.py's cpu_baseline* legs may import it).

Takes the same circuit dict as oracle/prove_ref.py (the Python restatement it is held against on small circuits,
tests/test_oracle_prove_c.py) and returns the proof in the reference's wire format."""
import ctypes

import numpy as np

from . import oracle as o

P = o.P

KINDS = {name: i for i, name in enumerate([
    "noop", "constant", "public_input", "arithmetic", "base_sum", "u32_add_many", "u32_arithmetic", "u32_subtraction", "u32_range_check",
    "comparison", "random_access", "poseidon", "arithmetic_extension", "mul_extension", "reducing", "reducing_extension", "exponentiation",
    "poseidon_mds", "low_degree_interpolation", "high_degree_interpolation"])}

_u64p = ctypes.POINTER(ctypes.c_uint64)
_u32p = ctypes.POINTER(ctypes.c_uint32)


class Gate(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_uint32), ("params", ctypes.c_uint32 * 3), ("selector_index", ctypes.c_uint32)]


class CircuitDesc(ctypes.Structure):
    _fields_ = [(k, ctypes.c_uint32) for k in ("degree_bits", "num_wires", "num_routed_wires", "num_constants", "num_challenges",
                                               "quotient_degree_factor", "num_gate_constraints", "rate_bits", "cap_height",
                                               "proof_of_work_bits", "num_query_rounds", "num_reductions")] + [
        ("reduction_arity_bits", _u32p), ("hiding", ctypes.c_uint32), ("k_is", _u64p), ("constants", _u64p), ("sigmas", _u64p),
        ("gates", ctypes.POINTER(Gate)), ("num_gates", ctypes.c_uint32), ("num_selectors", ctypes.c_uint32), ("group_bounds", _u32p),
        ("circuit_digest", _u64p)]


class Trace(ctypes.Structure):
    _fields_ = [(k, _u64p) for k in ("betas", "gammas", "alphas", "zeta", "zs_partial_products", "quotient_polys", "wires_cap", "zs_cap",
                                     "quotient_cap")] + [("stage_seconds", ctypes.c_double * 8)]


STAGES = ("wires commitment", "partial products", "zs commitment", "quotient polys", "quotient commitment", "opening set", "opening proof (FRI)", "total")

_bound = False


def lib():
    global _bound
    L = o.lib()
    if not _bound:
        L.glo_gate_constraints.restype = ctypes.c_int
        L.glo_gate_constraints.argtypes = [ctypes.POINTER(Gate), _u64p, _u64p, _u64p, _u64p]
        L.glo_gate_num_constraints.restype = ctypes.c_int
        L.glo_gate_num_constraints.argtypes = [ctypes.POINTER(Gate)]
        L.glo_evaluate_gate_constraints.restype = None
        L.glo_evaluate_gate_constraints.argtypes = [ctypes.POINTER(CircuitDesc), _u64p, _u64p, _u64p, _u64p]
        L.glo_circuit_new.restype = ctypes.c_void_p
        L.glo_circuit_new.argtypes = [ctypes.POINTER(CircuitDesc), ctypes.c_int]
        L.glo_circuit_free.restype = None
        L.glo_circuit_free.argtypes = [ctypes.c_void_p]
        L.glo_circuit_info.restype = None
        L.glo_circuit_info.argtypes = [ctypes.c_void_p, _u64p, _u64p]
        L.glo_prove.restype = ctypes.c_int
        L.glo_prove.argtypes = [ctypes.c_void_p, _u64p, _u64p, ctypes.c_uint32, _u64p, ctypes.POINTER(ctypes.POINTER(ctypes.c_uint8)),
                                ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(Trace), ctypes.c_int, ctypes.c_char_p, ctypes.c_size_t]
        L.glo_bytes_free.restype = None
        L.glo_bytes_free.argtypes = [ctypes.POINTER(ctypes.c_uint8)]
        _bound = True
    return L


def _params(param):
    if param is None:
        return (0, 0, 0)
    if isinstance(param, (tuple, list)):
        return tuple(int(x) for x in param) + (0,) * (3 - len(param))
    return (int(param), 0, 0)


def make_gate(kind, param, selector_index=0):
    g = Gate()
    g.kind = KINDS[kind]
    g.params[:] = _params(param)
    g.selector_index = selector_index
    return g


def gate_constraints(kind, param, consts, wires, pih):
    """Gate::eval_unfiltered_base_one for one point: list of constraint values (canonical)."""
    g = make_gate(kind, param)
    n = lib().glo_gate_num_constraints(ctypes.byref(g))
    assert n >= 0, kind
    c = o._arr(list(consts) + [0, 0])
    w = o._arr(list(wires) + [0] * 8)
    h = o._arr(pih)
    out = np.zeros(max(n, 1), dtype=np.uint64)
    got = lib().glo_gate_constraints(ctypes.byref(g), o._p(c), o._p(w), o._p(h), o._p(out))
    assert got == n, (kind, got, n)
    return [int(x) for x in o.canon(out[:n])]


def _mat(cols):
    a = np.ascontiguousarray(np.asarray(cols, dtype=np.uint64))
    assert a.ndim == 2
    return a


class Circuit:
    """The circuit with its preprocessed commitment (glo_circuit_new). `circuit` is the dict of oracle/prove_ref.py;
    its "constants_sigmas" / "circuit_digest" entries are not needed (pass derive_digest=False to take the dict's digest)."""

    def __init__(self, circuit, threads=None, derive_digest=True):
        self.threads = threads or o.usable_threads()  # the CPUs the container's quota grants, not the host's hardware threads
        fp = circuit["fri_params"]
        d = CircuitDesc()
        for k in ("degree_bits", "num_wires", "num_routed_wires", "num_constants", "num_challenges", "quotient_degree_factor", "num_gate_constraints"):
            setattr(d, k, int(circuit[k]))
        for k in ("rate_bits", "cap_height", "proof_of_work_bits", "num_query_rounds"):
            setattr(d, k, int(fp[k]))
        ab = np.array(list(fp["reduction_arity_bits"]) + [0], dtype=np.uint32)
        d.num_reductions = len(fp["reduction_arity_bits"])
        d.reduction_arity_bits = ab.ctypes.data_as(_u32p)
        d.hiding = 1 if fp.get("hiding") else 0
        k_is = o._arr([int(x) % P for x in circuit["k_is"]])
        consts, sig = _mat(circuit["constants"]), _mat(circuit["sigmas"])
        n = 1 << d.degree_bits
        assert consts.shape == (d.num_constants, n) and sig.shape == (d.num_routed_wires, n)
        gates = (Gate * len(circuit["gates"]))()
        for i, (kind, param) in enumerate(circuit["gates"]):
            gates[i] = make_gate(kind, param, int(circuit["selector_indices"][i]))
        gb = np.array([x for g in circuit["groups"] for x in g], dtype=np.uint32)
        d.k_is, d.constants, d.sigmas = o._p(k_is), o._p(consts), o._p(sig)
        d.gates, d.num_gates, d.num_selectors = gates, len(circuit["gates"]), len(circuit["groups"])
        d.group_bounds = gb.ctypes.data_as(_u32p)
        dig = None
        if not derive_digest:
            dig = o._arr([int(x) for x in circuit["circuit_digest"]])
            d.circuit_digest = o._p(dig)
        self.desc, self._keep = d, (ab, k_is, consts, sig, gates, gb, dig)
        self.handle = lib().glo_circuit_new(ctypes.byref(d), self.threads)
        if not self.handle:
            raise ValueError("glo_circuit_new failed")
        self.num_challenges, self.degree_bits, self.cap_height = d.num_challenges, d.degree_bits, d.cap_height
        self.qdf, self.num_routed, self.rate_bits, self.num_wires = d.quotient_degree_factor, d.num_routed_wires, d.rate_bits, d.num_wires
        dg = np.zeros(4, dtype=np.uint64)
        cap = np.zeros((1 << d.cap_height, 4), dtype=np.uint64)
        lib().glo_circuit_info(self.handle, o._p(dg), o._p(cap))
        self.circuit_digest = [int(x) for x in dg]
        self.constants_sigmas_cap = [[int(x) for x in h] for h in cap]

    def evaluate_gate_constraints(self, local_constants, local_wires, pih):
        out = np.zeros(max(self.desc.num_gate_constraints, 1), dtype=np.uint64)
        lc, lw, h = o._arr(local_constants), o._arr(list(local_wires) + [0] * 8), o._arr(pih)
        lib().glo_evaluate_gate_constraints(ctypes.byref(self.desc), o._p(lc), o._p(lw), o._p(h), o._p(out))
        return [int(x) for x in o.canon(out[: self.desc.num_gate_constraints])]

    def prove(self, wires, public_inputs, salts=None, trace=None):
        """Proof bytes. `trace` (a dict) receives challenges, Z / partial-product values, quotient polynomials, caps, stage times."""
        w = _mat(wires)
        n = 1 << self.degree_bits
        assert w.shape == (self.num_wires, n), w.shape
        pis = o._arr([int(x) % P for x in public_inputs] + [0])
        s = None
        if salts is not None:
            s = np.ascontiguousarray(np.asarray(salts, dtype=np.uint64))
            assert s.shape == (3, 4, n << self.rate_bits)
        t = Trace()
        keep = {}
        if trace is not None:
            nch, qdb = self.num_challenges, (self.qdf - 1).bit_length()
            num_prods = -(-self.num_routed // self.qdf) - 1
            shapes = dict(betas=(nch,), gammas=(nch,), alphas=(nch,), zeta=(2,), zs_partial_products=(nch * (1 + num_prods), n),
                          quotient_polys=(nch, n << qdb), wires_cap=(1 << self.cap_height, 4), zs_cap=(1 << self.cap_height, 4),
                          quotient_cap=(1 << self.cap_height, 4))
            for k, shp in shapes.items():
                keep[k] = np.zeros(shp, dtype=np.uint64)
                setattr(t, k, o._p(keep[k]))
        out = ctypes.POINTER(ctypes.c_uint8)()
        out_len = ctypes.c_size_t()
        err = ctypes.create_string_buffer(256)
        rc = lib().glo_prove(self.handle, o._p(w), o._p(pis), len(public_inputs), o._p(s) if s is not None else None, ctypes.byref(out),
                             ctypes.byref(out_len), ctypes.byref(t), self.threads, err, 256)
        if rc != 0:
            raise AssertionError(err.value.decode())
        data = ctypes.string_at(out, out_len.value)
        lib().glo_bytes_free(out)
        if trace is not None:
            trace.update({k: o.canon(v) for k, v in keep.items()})
            trace["stage_seconds"] = dict(zip(STAGES, t.stage_seconds))
        return data

    def close(self):
        if self.handle:
            lib().glo_circuit_free(self.handle)
            self.handle = None

    def __del__(self):
        self.close()


def prove(circuit, wires, public_inputs, salts=None, threads=None, trace=None):
    c = Circuit(circuit, threads=threads)
    try:
        return c.prove(wires, public_inputs, salts=salts, trace=trace)
    finally:
        c.close()
