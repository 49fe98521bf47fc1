"""Host-side mirror of plonky2/src/plonk/prover.rs: prove() and the stages it calls."""
import ctypes

import numpy as np

from . import _lib
from .device import DeviceBuffer

COSET_SHIFT = 7


def num_partial_products(n, max_degree):
    """plonky2/src/util/partial_products.rs:41-48"""
    return -(-n // max_degree) - 1


def _host_u64(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


def all_wires_permutation_partial_products(ctx, d_wires, wires_stride, d_sigmas, sigmas_stride, d_k_is, betas, gammas, num_routed,
                                           quotient_degree_factor, log_n):
    """all_wires_permutation_partial_products (prover.rs:702-723) followed by the reordering of
    prover.rs:112-117: returns a DeviceBuffer [num_challenges*(1+num_prods)][n] (Z first), ready to be
    committed with PolynomialBatch.from_values_device."""
    b, g = _host_u64(betas), _host_u64(gammas)
    assert b.size == g.size
    n_cols = b.size * (1 + num_partial_products(num_routed, quotient_degree_factor))
    out = DeviceBuffer(ctx, n_cols << log_n)
    _lib.call("gl_permutation_partial_products", d_wires.ptr, wires_stride, d_sigmas.ptr, sigmas_stride, d_k_is.ptr,
              b.ctypes.data, g.ctypes.data, b.size, num_routed, quotient_degree_factor, log_n, out.ptr, ctx.ptr)
    return out, n_cols


def compute_quotient_polys(ctx, wires_commitment, constants_sigmas_commitment, zs_partial_products_commitment, num_constants,
                           num_routed, d_k_is, betas, gammas, alphas, quotient_degree_factor, d_gate_terms=None,
                           num_gate_constraints=0, gate_program=None, column_major=True, public_inputs_hash=None):
    """compute_quotient_polys (prover.rs:790-1034) over three PolynomialBatch commitments resident in
    HBM, read from their column-major LDE (coalesced; `column_major=False` reads the leaf-major rows
    like get_lde_values). Gate constraints: `d_gate_terms` (a term array), or `gate_program` — run by
    its run-time compiled kernel when `gate_program.compile()` has been called, else by the
    interpreter. Returns a DeviceBuffer of coefficients [num_challenges][n << qdb]."""
    b, g, a = _host_u64(betas), _host_u64(gammas), _host_u64(alphas)
    wc, cc, zc = wires_commitment, constants_sigmas_commitment, zs_partial_products_commitment
    qdb = (quotient_degree_factor - 1).bit_length()
    n_ext = 1 << (wc.degree_log + wc.rate_bits)
    if column_major:
        ptrs = (wc.d_lde.ptr, cc.d_lde.ptr, zc.d_lde.ptr)
    else:
        ptrs = (wc.merkle_tree.d_leaves.ptr, cc.merkle_tree.d_leaves.ptr, zc.merkle_tree.d_leaves.ptr)
    kernel = gate_program.kernel if gate_program is not None else None
    pih = _host_u64(public_inputs_hash if public_inputs_hash is not None else (gate_program.public_inputs_hash if gate_program else [0] * 4))
    work = DeviceBuffer(ctx, b.size << (wc.degree_log + qdb)) if kernel else None
    args = _lib.GlQuotientArgs(
        ptrs[0], ptrs[1], ptrs[2],
        wc.merkle_tree.leaf_len, cc.merkle_tree.leaf_len, zc.merkle_tree.leaf_len,
        d_k_is.ptr, d_gate_terms.ptr if d_gate_terms is not None else None,
        b.ctypes.data, g.ctypes.data, a.ctypes.data,
        num_constants, num_routed, b.size, num_gate_constraints,
        wc.degree_log, wc.rate_bits, quotient_degree_factor, COSET_SHIFT,
        ctypes.pointer(gate_program.struct) if (gate_program is not None and not kernel) else None,
        n_ext if column_major else 0, kernel, pih.ctypes.data if kernel else None, work.ptr if kernel else None,
    )
    out = DeviceBuffer(ctx, b.size << (wc.degree_log + qdb))
    _lib.call("gl_compute_quotient_polys", ctypes.byref(args), out.ptr, ctx.ptr)
    if work is not None:
        ctx.synchronize()
        work.free()
    return out


def reference_compute_quotient_polys(ctx, d_wires_leaves, log_len, d_zs_partial_products_leaves, d_constants_sigmas_leaves, d_k_is,
                                     d_alphas, d_betas, d_gammas, salt_size=0):
    """The reference's own FFI symbol `compute_quotient_polys` (cuda/src/lib.rs:117-143, called at
    plonky2/src/plonk/prover.rs:539-566): the ed25519 circuit is compiled into the library (ed25519_circuit.py),
    every buffer is a DEVICE buffer — three leaf-major LDEs [n_ext][234 + salt | 20 | 88], k_is (80) and the three
    challenge pairs. Returns the DeviceBuffer d_quotient_polys [2][n_ext] (coefficients)."""
    from . import ed25519_circuit as ed

    n = 1 << log_len
    n_ext = n << ed.RATE_BITS
    sl = lambda buf, count: _lib.GlDataSlice(buf.ptr, count)  # noqa: E731
    zs = sl(d_zs_partial_products_leaves, n_ext * ed.ZS_PARTIAL_PRODUCTS_LEAF_LEN)
    cs = sl(d_constants_sigmas_leaves, n_ext * ed.CONSTANTS_SIGMAS_LEAF_LEN)
    k_is, al, be, ga = sl(d_k_is, ed.NUM_ROUTED_WIRES), sl(d_alphas, 2), sl(d_betas, 2), sl(d_gammas, 2)
    d_outs = DeviceBuffer(ctx, 2 * n_ext)
    out = DeviceBuffer(ctx, 2 * n_ext)
    ref = ctypes.addressof
    try:
        _lib.call("compute_quotient_polys", d_wires_leaves.ptr, ed.NUM_WIRES, n, log_len, None, None, ed.RATE_BITS, salt_size,
                  ref(zs), ref(cs), d_outs.ptr, out.ptr, None, None, None, ref(k_is), ref(al), ref(be), ref(ga), ctx.ptr)
    finally:
        d_outs.free()
    return out


def reference_set_public_inputs_hash(public_inputs_hash=None):
    """Replace (None: restore) the public-inputs hash the reference compiles into its kernel (plonky2_gpu.cu:686-689)."""
    _lib.call("gl_reference_set_public_inputs_hash", _host_u64(public_inputs_hash) if public_inputs_hash is not None else None)


class GateProgram:
    """Device-resident gate programs of a circuit (see gate_program.py): the table-driven replacement
    of the reference's hard-wired gate list (cuda/plonky2_gpu_impl.cuh:600-685)."""

    def __init__(self, ctx, gate_instrs, selector_indices, groups, public_inputs_hash, immediates=None):
        from . import gate_program as gp

        instrs, descs = gp.pack_program(gate_instrs, selector_indices, groups)
        self._instrs, self._descs = np.ascontiguousarray(instrs), np.ascontiguousarray(descs)
        self._imms = _host_u64(immediates) if immediates is not None else None
        self.num_selectors = len(groups)
        self.kernel = None
        self.public_inputs_hash = [int(v) for v in public_inputs_hash]
        self.d_instrs = DeviceBuffer.from_host(ctx, np.frombuffer(np.ascontiguousarray(instrs).tobytes(), dtype=np.uint64))
        d32 = np.ascontiguousarray(descs).tobytes()
        d32 += b"\0" * (-len(d32) % 8)
        self.d_gates = DeviceBuffer.from_host(ctx, np.frombuffer(d32, dtype=np.uint64))
        self.d_imms = DeviceBuffer.from_host(ctx, _host_u64(immediates)) if immediates is not None else None
        self.struct = _lib.GlGateProgram(
            self.d_instrs.ptr, self.d_gates.ptr, self.d_imms.ptr if self.d_imms else None, len(gate_instrs), len(groups),
            (ctypes.c_uint64 * 4)(*[int(v) for v in public_inputs_hash]),
        )

    def set_public_inputs_hash(self, pih):
        """the hash is per proof (prover.rs:52), the programs per circuit"""
        self.public_inputs_hash = [int(v) for v in pih]
        self.struct.public_inputs_hash = (ctypes.c_uint64 * 4)(*self.public_inputs_hash)

    def compile(self, num_gate_constraints, num_challenges):
        """Turn the programs into a kernel specialised to this circuit (hiprtc, once per circuit)."""
        k = ctypes.c_void_p()
        n_instr = self._instrs.size // 4
        if self.kernel:
            _lib.load().gl_gate_kernel_destroy(self.kernel)
            self.kernel = None
        _lib.call("gl_gate_kernel_build", self._instrs, n_instr, self._descs, self._descs.size // 6, self._imms,
                  0 if self._imms is None else self._imms.size, self.num_selectors, num_gate_constraints, num_challenges, ctypes.byref(k))
        self.kernel = k.value
        return self

    def kernel_source(self):
        return _lib.load().gl_gate_kernel_source(self.kernel).decode() if self.kernel else ""

    def __del__(self):
        try:
            if self.kernel:
                _lib.load().gl_gate_kernel_destroy(self.kernel)
                self.kernel = None
        except Exception:
            pass


class CircuitData:
    """What prove() needs of CommonCircuitData + ProverOnlyCircuitData (plonk/circuit_data.rs), resident
    in HBM: the preprocessed constants_sigmas commitment, the sigma value columns, k_is and the gate
    programs. `circuit` is the plain dict described in INTEGRATION.md section 7."""

    def __init__(self, ctx, circuit, compile_gates=True):
        from . import gate_program as gp
        from .polynomial_batch import PolynomialBatch

        self.ctx = ctx
        for k in ("degree_bits", "num_wires", "num_routed_wires", "num_constants", "num_challenges", "quotient_degree_factor",
                  "num_gate_constraints", "fri_params", "circuit_digest"):
            setattr(self, k, circuit[k])
        fp = self.fri_params
        cs = np.ascontiguousarray(list(circuit["constants"]) + list(circuit["sigmas"]), dtype=np.uint64)
        self.constants_sigmas_commitment = PolynomialBatch.from_values(ctx, cs, fp["rate_bits"], False, fp["cap_height"], leaf_major=False)
        self.d_sigmas = DeviceBuffer.from_host(ctx, _host_u64(circuit["sigmas"]))
        self.d_k_is = DeviceBuffer.from_host(ctx, _host_u64(circuit["k_is"]))
        pool = gp.ImmediatePool()
        programs = [gp.build_gate(kind, param, pool) for kind, param in circuit["gates"]]
        self.gate_program = GateProgram(ctx, programs, circuit["selector_indices"], circuit["groups"], [0, 0, 0, 0],
                                        immediates=pool.values or None)
        if compile_gates:
            self.gate_program.compile(self.num_gate_constraints, self.num_challenges)

    def fri_instance(self, zeta):
        """get_fri_instance (plonk/circuit_data.rs:351-371)"""
        from .fri import ext_mul

        nc = self.num_challenges
        counts = [self.num_constants + self.num_routed_wires, self.num_wires,
                  nc * (1 + num_partial_products(self.num_routed_wires, self.quotient_degree_factor)), nc * self.quotient_degree_factor]
        all_polys = [(oi, pi) for oi, k in enumerate(counts) for pi in range(k)]
        g = pow(ROOT_OF_UNITY_2_32, 1 << (32 - self.degree_bits), P)
        return dict(batches=[(zeta, all_polys), (ext_mul((g, 0), zeta), [(2, i) for i in range(nc)])])


P = 0xFFFFFFFF00000001
ROOT_OF_UNITY_2_32 = 1753635133440165772  # field/src/goldilocks_field.rs:89


def _pairs(a):
    return [(int(x[0]), int(x[1])) for x in a]


def prove(ctx, cd, wires, public_inputs, timing=None):
    """prove() (plonk/prover.rs:40-233) from the full witness on: `wires` is the [num_wires][n] matrix of
    wire values (host array, or a DeviceBuffer that is left untouched). Every polynomial, LDE and tree
    stays in HBM; the host sees caps, challenges, openings and the FRI proof."""
    import time

    from . import fri
    from .challenger import Challenger, hash_no_pad
    from .polynomial_batch import PolynomialBatch

    def stage(name, t0):
        if timing is not None:
            ctx.synchronize()
            timing[name] = timing.get(name, 0.0) + (time.perf_counter() - t0) * 1e3
        return time.perf_counter()

    fp = cd.fri_params
    rate_bits, cap_height = fp["rate_bits"], fp["cap_height"]
    db, n = cd.degree_bits, 1 << cd.degree_bits
    nch, qdf, num_routed = cd.num_challenges, cd.quotient_degree_factor, cd.num_routed_wires
    t = time.perf_counter()
    pih = hash_no_pad(ctx, public_inputs)
    d_wire_values = wires if isinstance(wires, DeviceBuffer) else DeviceBuffer.from_host(ctx, _host_u64(wires))
    d_w = DeviceBuffer(ctx, cd.num_wires * n)
    _lib.call("gl_memcpy_d2d", d_w.ptr, d_wire_values.ptr, cd.num_wires * n * 8, ctx.ptr)
    t = stage("upload witness", t)
    wires_c = PolynomialBatch.from_values_device(ctx, d_w, cd.num_wires, db, rate_bits, False, cap_height, leaf_major=False)
    t = stage("wires commitment", t)
    ch = Challenger(ctx)
    ch.observe_hash(cd.circuit_digest)
    ch.observe_hash(pih)
    ch.observe_cap(wires_c.merkle_tree.cap.tolist())
    betas, gammas = ch.get_n_challenges(nch), ch.get_n_challenges(nch)
    if not qdf < num_routed:
        raise ValueError("When the number of routed wires is smaller that the degree, we should change the logic to avoid "
                         "computing partial products.")
    d_zpp, n_cols = all_wires_permutation_partial_products(ctx, d_wire_values, n, cd.d_sigmas, n, cd.d_k_is, betas, gammas, num_routed,
                                                           qdf, db)
    t = stage("partial products", t)
    zs_c = PolynomialBatch.from_values_device(ctx, d_zpp, n_cols, db, rate_bits, False, cap_height, leaf_major=False)
    t = stage("zs partial products commitment", t)
    ch.observe_cap(zs_c.merkle_tree.cap.tolist())
    alphas = ch.get_n_challenges(nch)
    cd.gate_program.set_public_inputs_hash(pih)
    d_q = compute_quotient_polys(ctx, wires_c, cd.constants_sigmas_commitment, zs_c, cd.num_constants, num_routed, cd.d_k_is, betas,
                                 gammas, alphas, qdf, None, cd.num_gate_constraints, cd.gate_program)
    t = stage("quotient polys", t)
    qdb = (qdf - 1).bit_length()
    if qdf == 1 << qdb:
        d_chunks = d_q  # [nch][n << qdb] read flat is already [nch * qdf][n]
    else:
        d_chunks = DeviceBuffer(ctx, nch * qdf * n)
        for c in range(nch):
            tail = d_q.download((c << (db + qdb)) + qdf * n, (n << qdb) - qdf * n)
            if tail.any():
                raise ValueError("Quotient has failed, the vanishing polynomial is not divisible by Z_H")
            _lib.call("gl_memcpy_d2d", d_chunks.ptr + 8 * c * qdf * n, d_q.ptr + 8 * (c << (db + qdb)), 8 * qdf * n, ctx.ptr)
    quot_c = PolynomialBatch.from_coeffs_device(ctx, d_chunks, nch * qdf, db, rate_bits, False, cap_height, leaf_major=False)
    t = stage("quotient commitment", t)
    ch.observe_cap(quot_c.merkle_tree.cap.tolist())
    zeta = ch.get_extension_challenge()
    if fri.ext_pow(zeta, n) == (1, 0):
        raise ValueError("Opening point is in the subgroup.")
    g = pow(ROOT_OF_UNITY_2_32, 1 << (32 - db), P)
    g_zeta = fri.ext_mul((g, 0), zeta)
    cs_eval = _pairs(cd.constants_sigmas_commitment.eval_polynomials_ext2([zeta])[0])
    zs_evals = zs_c.eval_polynomials_ext2([zeta, g_zeta])
    zs_eval, zs_next = _pairs(zs_evals[0]), _pairs(zs_evals[1])
    openings = dict(constants=cs_eval[: cd.num_constants], plonk_sigmas=cs_eval[cd.num_constants :],
                    wires=_pairs(wires_c.eval_polynomials_ext2([zeta])[0]), plonk_zs=zs_eval[:nch], plonk_zs_next=zs_next[:nch],
                    partial_products=zs_eval[nch:], quotient_polys=_pairs(quot_c.eval_polynomials_ext2([zeta])[0]))
    t = stage("opening set", t)
    # OpeningSet::to_fri_openings (plonk/proof.rs:336-356)
    ch.observe_extension_elements(openings["constants"] + openings["plonk_sigmas"] + openings["wires"] + openings["plonk_zs"]
                                  + openings["partial_products"] + openings["quotient_polys"])
    ch.observe_extension_elements(openings["plonk_zs_next"])
    opening_proof = fri.prove_openings(ctx, cd.fri_instance(zeta), [cd.constants_sigmas_commitment, wires_c, zs_c, quot_c], ch, fp, timing)
    stage("opening proof (FRI)", t)
    return dict(wires_cap=wires_c.merkle_tree.cap.tolist(), plonk_zs_partial_products_cap=zs_c.merkle_tree.cap.tolist(),
                quotient_polys_cap=quot_c.merkle_tree.cap.tolist(), openings=openings, opening_proof=opening_proof,
                public_inputs=[int(x) for x in public_inputs])


class NativeCircuit:
    """The circuit object of the library's own prover (gl_circuit_create): same input dict as
    CircuitData, but the whole of prove() then runs inside one native call (gl_prove, csrc/prove.hip)."""

    def __init__(self, ctx, circuit, compile_gates=True):
        from . import gate_program as gp

        self.ctx = ctx
        self.circuit = circuit
        pool = gp.ImmediatePool()
        programs = [gp.build_gate(kind, param, pool) for kind, param in circuit["gates"]]
        instrs, descs = gp.pack_program(programs, circuit["selector_indices"], circuit["groups"])
        instrs, descs = np.ascontiguousarray(instrs), np.ascontiguousarray(descs)
        imms = _host_u64(pool.values) if pool.values else None
        fp = circuit["fri_params"]
        arity = np.ascontiguousarray(fp["reduction_arity_bits"], dtype=np.uint32)
        k_is, consts, sigmas = _host_u64(circuit["k_is"]), _host_u64(circuit["constants"]), _host_u64(circuit["sigmas"])
        digest = _host_u64(circuit["circuit_digest"]) if circuit.get("circuit_digest") is not None else None
        desc = _lib.GlCircuitDesc(
            ctypes.sizeof(_lib.GlCircuitDesc), circuit["degree_bits"], circuit["num_wires"], circuit["num_routed_wires"], circuit["num_constants"], circuit["num_challenges"],
            circuit["quotient_degree_factor"], circuit["num_gate_constraints"],
            _lib.GlFriParams(fp["rate_bits"], fp["cap_height"], fp["proof_of_work_bits"], fp["num_query_rounds"], arity.size,
                             arity.ctypes.data, 1 if fp.get("hiding") else 0),
            k_is.ctypes.data, consts.ctypes.data, sigmas.ctypes.data,
            instrs.ctypes.data, instrs.size // 4, descs.ctypes.data, descs.size // 6,
            imms.ctypes.data if imms is not None else None, 0 if imms is None else imms.size, len(circuit["groups"]),
            1 if compile_gates else 0, digest.ctypes.data if digest is not None else None)
        h = ctypes.c_void_p()
        _lib.call("gl_circuit_create", ctypes.byref(desc), ctypes.byref(h), ctx.ptr)
        self.ptr = h.value
        dg = np.zeros(4, dtype=np.uint64)
        cap = np.zeros(4 << fp["cap_height"], dtype=np.uint64)
        _lib.call("gl_circuit_info", self.ptr, dg, cap)
        self.circuit_digest = [int(x) for x in dg]
        self.constants_sigmas_cap = cap.reshape(-1, 4).tolist()

    def prove_bytes(self, wires, public_inputs, timing=None, salts=None, ctx=None):
        """gl_prove: the proof in the reference's wire format. `wires`: host [num_wires][n] or a DeviceBuffer.
        `ctx`: prove on another context than the one the circuit was created with (same device) — several host threads, each
        with its own context, may prove with one circuit handle at the same time (the handle keeps a buffer pool per context).
        `salts` (a circuit with fri_params["hiding"], i.e. zero_knowledge): [3][4][n_ext] uniform field elements, host or DeviceBuffer —
        the blinding of the wires, Zs / partial products and quotient commitments in leaf order (gl_prove_zk)."""
        ctx = ctx or self.ctx
        d_w = wires if isinstance(wires, DeviceBuffer) else DeviceBuffer.from_host(ctx, _host_u64(wires))
        pis = _host_u64(public_inputs)
        out, ln = ctypes.c_void_p(), ctypes.c_uint64()
        ms = np.zeros(_lib.GL_PROVE_STAGES, dtype=np.float64) if timing is not None else None
        if salts is not None:
            d_s = salts if isinstance(salts, DeviceBuffer) else DeviceBuffer.from_host(ctx, _host_u64(salts))
            _lib.call("gl_prove_zk", self.ptr, d_w.ptr, pis, pis.size, d_s.ptr, ctypes.byref(out), ctypes.byref(ln), ms, ctx.ptr)
        else:
            _lib.call("gl_prove", self.ptr, d_w.ptr, pis, pis.size, ctypes.byref(out), ctypes.byref(ln), ms, ctx.ptr)
        data = ctypes.string_at(out.value, ln.value)
        _lib.load().gl_bytes_free(out.value)
        if timing is not None:
            for name, v in zip(_lib.PROVE_STAGE_NAMES, ms):
                timing[name] = timing.get(name, 0.0) + float(v)
        return data

    def prove_many(self, wires_list, public_inputs_list, ctxs):
        """gl_prove_many: one proof per entry of `wires_list` (DeviceBuffers on this circuit's device), len(ctxs) of them in flight —
        worker w proves entries w, w + len(ctxs), .. on ctxs[w]. Returns the proofs' bytes in order."""
        count, k = len(wires_list), len(ctxs)
        pis = [_host_u64(p) for p in public_inputs_list]
        npi = pis[0].size if pis else 0
        assert all(p.size == npi for p in pis) and len(pis) == count
        d_w = (ctypes.c_void_p * max(count, 1))(*[w.ptr for w in wires_list])
        h_p = (ctypes.c_void_p * max(count, 1))(*[p.ctypes.data for p in pis])
        outs, lens = (ctypes.c_void_p * max(count, 1))(), (ctypes.c_uint64 * max(count, 1))()
        cx = (ctypes.c_void_p * max(k, 1))(*[c.ptr for c in ctxs])
        _lib.call("gl_prove_many", self.ptr, ctypes.addressof(d_w), ctypes.addressof(h_p), npi, count, ctypes.addressof(outs), ctypes.addressof(lens),
                  ctypes.addressof(cx), k)
        proofs = []
        for i in range(count):
            proofs.append(ctypes.string_at(outs[i], lens[i]))
            _lib.load().gl_bytes_free(outs[i])
        return proofs

    def prove(self, wires, public_inputs, timing=None, salts=None):
        from . import serialization

        return serialization.proof_from_bytes(self.prove_bytes(wires, public_inputs, timing, salts), self.circuit)

    def trim(self):
        """release the working buffers gl_prove keeps attached to the circuit between proofs"""
        _lib.call("gl_circuit_trim", self.ptr)

    def close(self):
        if self.ptr:
            _lib.load().gl_circuit_destroy(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
