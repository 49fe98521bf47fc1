"""Host-side mirror of the permutation-argument steps of plonky2/src/plonk/prover.rs."""
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
                           num_gate_constraints=0, gate_program=None):
    """compute_quotient_polys (prover.rs:790-1034) over three PolynomialBatch commitments (leaf-major
    leaves resident in HBM). Returns a DeviceBuffer of coefficients [num_challenges][n << qdb]."""
    b, g, a = _host_u64(betas), _host_u64(gammas), _host_u64(alphas)
    wc, cc, zc = wires_commitment, constants_sigmas_commitment, zs_partial_products_commitment
    qdb = (quotient_degree_factor - 1).bit_length()
    args = _lib.GlQuotientArgs(
        wc.merkle_tree.d_leaves.ptr, cc.merkle_tree.d_leaves.ptr, zc.merkle_tree.d_leaves.ptr,
        wc.merkle_tree.leaf_len, cc.merkle_tree.leaf_len, zc.merkle_tree.leaf_len,
        d_k_is.ptr, d_gate_terms.ptr if d_gate_terms is not None else None,
        b.ctypes.data, g.ctypes.data, a.ctypes.data,
        num_constants, num_routed, b.size, num_gate_constraints,
        wc.degree_log, wc.rate_bits, quotient_degree_factor, COSET_SHIFT,
        ctypes.pointer(gate_program.struct) if gate_program is not None else None,
    )
    out = DeviceBuffer(ctx, b.size << (wc.degree_log + qdb))
    _lib.call("gl_compute_quotient_polys", ctypes.byref(args), out.ptr, ctx.ptr)
    return out


class GateProgram:
    """Device-resident gate programs of a circuit (see gate_program.py): the table-driven replacement
    of the reference's hard-wired gate list (cuda/plonky2_gpu_impl.cuh:600-685)."""

    def __init__(self, ctx, gate_instrs, selector_indices, groups, public_inputs_hash, immediates=None):
        from . import gate_program as gp

        instrs, descs = gp.pack_program(gate_instrs, selector_indices, groups)
        self.d_instrs = DeviceBuffer.from_host(ctx, np.frombuffer(np.ascontiguousarray(instrs).tobytes(), dtype=np.uint64))
        d32 = np.ascontiguousarray(descs).tobytes()
        d32 += b"\0" * (-len(d32) % 8)
        self.d_gates = DeviceBuffer.from_host(ctx, np.frombuffer(d32, dtype=np.uint64))
        self.d_imms = DeviceBuffer.from_host(ctx, _host_u64(immediates)) if immediates is not None else None
        self.struct = _lib.GlGateProgram(
            self.d_instrs.ptr, self.d_gates.ptr, self.d_imms.ptr if self.d_imms else None, len(gate_instrs), len(groups),
            (ctypes.c_uint64 * 4)(*[int(v) for v in public_inputs_hash]),
        )
