"""ctypes binding of libplonky2_hip.so (the C ABI in include/plonky2_hip.h).

There is no CPU fallback: if the HIP library is missing or a call fails, this raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# PLONKY2_HIP_LIBRARY=<path>: load another build of the library (diagnostic builds, A/B measurements); no fallback either way
LIB_PATH = os.environ.get("PLONKY2_HIP_LIBRARY") or os.path.join(_HERE, "libplonky2_hip.so")


class GlError(ctypes.Structure):
    _fields_ = [("code", ctypes.c_int), ("message", ctypes.c_void_p)]


class GlDataSlice(ctypes.Structure):
    _fields_ = [("ptr", ctypes.c_void_p), ("len", ctypes.c_int)]


class GlObserveSrc(ctypes.Structure):
    _fields_ = [("d_ptr", ctypes.c_void_p), ("count", ctypes.c_uint64), ("planar_len", ctypes.c_uint64)]


class GlGateProgram(ctypes.Structure):
    _fields_ = [
        ("d_instrs", ctypes.c_void_p),
        ("d_gates", ctypes.c_void_p),
        ("d_immediates", ctypes.c_void_p),
        ("num_gates", ctypes.c_uint32),
        ("num_selectors", ctypes.c_uint32),
        ("public_inputs_hash", ctypes.c_uint64 * 4),
    ]


class GlQuotientArgs(ctypes.Structure):
    _fields_ = [
        ("d_wires_leaves", ctypes.c_void_p),
        ("d_constants_sigmas_leaves", ctypes.c_void_p),
        ("d_zs_partial_products_leaves", ctypes.c_void_p),
        ("wires_leaf_len", ctypes.c_uint32),
        ("constants_sigmas_leaf_len", ctypes.c_uint32),
        ("zs_partial_products_leaf_len", ctypes.c_uint32),
        ("d_k_is", ctypes.c_void_p),
        ("d_gate_constraint_terms", ctypes.c_void_p),
        ("h_betas", ctypes.c_void_p),
        ("h_gammas", ctypes.c_void_p),
        ("h_alphas", ctypes.c_void_p),
        ("num_constants", ctypes.c_uint32),
        ("num_routed_wires", ctypes.c_uint32),
        ("num_challenges", ctypes.c_uint32),
        ("num_gate_constraints", ctypes.c_uint32),
        ("degree_bits", ctypes.c_uint32),
        ("rate_bits", ctypes.c_uint32),
        ("quotient_degree_factor", ctypes.c_uint32),
        ("coset_shift", ctypes.c_uint64),
        ("gate_program", ctypes.POINTER(GlGateProgram)),
        ("column_stride", ctypes.c_uint64),
        ("gate_kernel", ctypes.c_void_p),
        ("h_public_inputs_hash", ctypes.c_void_p),
        ("d_gate_workspace", ctypes.c_void_p),
    ]


class GlGateSpec(ctypes.Structure):
    _fields_ = [("kind", ctypes.c_uint32), ("params", ctypes.c_uint32 * 3), ("selector_index", ctypes.c_uint32)]


class GlGatePrograms(ctypes.Structure):
    _fields_ = [("instrs", ctypes.c_void_p), ("gates", ctypes.c_void_p), ("immediates", ctypes.c_void_p), ("num_instrs", ctypes.c_uint32),
                ("num_gates", ctypes.c_uint32), ("num_immediates", ctypes.c_uint32), ("num_gate_constraints", ctypes.c_uint32)]


# include/plonky2_hip.h enum GlGateKind, by the names plonky2_gpu_amd/gate_program.py uses
GATE_KINDS = {"noop": 0, "constant": 1, "public_input": 2, "arithmetic": 3, "base_sum": 4, "u32_add_many": 5, "u32_arithmetic": 6,
              "u32_subtraction": 7, "u32_range_check": 8, "comparison": 9, "random_access": 10, "poseidon": 11,
              "arithmetic_extension": 12, "mul_extension": 13, "reducing": 14, "reducing_extension": 15, "exponentiation": 16,
              "poseidon_mds": 17, "low_degree_interpolation": 18, "high_degree_interpolation": 19}


class GlFriParams(ctypes.Structure):
    _fields_ = [
        ("rate_bits", ctypes.c_uint32),
        ("cap_height", ctypes.c_uint32),
        ("proof_of_work_bits", ctypes.c_uint32),
        ("num_query_rounds", ctypes.c_uint32),
        ("num_reductions", ctypes.c_uint32),
        ("reduction_arity_bits", ctypes.c_void_p),
        ("hiding", ctypes.c_uint32),
    ]


class GlCircuitDesc(ctypes.Structure):
    _fields_ = [
        ("struct_size", ctypes.c_uint32),
        ("degree_bits", ctypes.c_uint32),
        ("num_wires", ctypes.c_uint32),
        ("num_routed_wires", ctypes.c_uint32),
        ("num_constants", ctypes.c_uint32),
        ("num_challenges", ctypes.c_uint32),
        ("quotient_degree_factor", ctypes.c_uint32),
        ("num_gate_constraints", ctypes.c_uint32),
        ("fri", GlFriParams),
        ("h_k_is", ctypes.c_void_p),
        ("h_constants", ctypes.c_void_p),
        ("h_sigmas", ctypes.c_void_p),
        ("h_instrs", ctypes.c_void_p),
        ("num_instrs", ctypes.c_uint32),
        ("h_gates", ctypes.c_void_p),
        ("num_gates", ctypes.c_uint32),
        ("h_immediates", ctypes.c_void_p),
        ("num_immediates", ctypes.c_uint32),
        ("num_selectors", ctypes.c_uint32),
        ("compile_gates", ctypes.c_int),
        ("h_circuit_digest", ctypes.c_void_p),
    ]


GL_PROVE_STAGES = 11
PROVE_STAGE_NAMES = ["wires commitment", "partial products", "zs partial products commitment", "quotient polys", "quotient commitment",
                     "opening set", "fri: combine + divide", "fri: commit phase", "fri: proof of work", "fri: query rounds", "serialise"]


class Plonky2HipError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"plonky2_hip error {code}: {message}")
        self.code = code


GL_E_INVALID = -1
GL_E_UNSUPPORTED = -2

_vp, _u64, _u32, _i = ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int

# name -> (restype, argtypes); every symbol declared in include/plonky2_hip.h
SIGNATURES = {
    "gl_version": (ctypes.c_char_p, []),
    "gl_device_count": (_i, []),
    "gl_ctx_create": (_vp, [_i]),
    "gl_ctx_destroy": (None, [_vp]),
    "gl_ctx_synchronize": (GlError, [_vp]),
    "gl_ctx_release": (None, [_vp]),
    "gl_workspace_bytes": (_u64, []),
    "gl_ctx_set_workspace": (GlError, [_vp, _vp, _u64]),
    "gl_pack_leaf_ranges": (GlError, [_vp, _u64, _u32, _u64, _u32, _vp, _vp]),
    "gl_malloc": (GlError, [ctypes.POINTER(_vp), _u64]),
    "gl_ctx_malloc": (GlError, [ctypes.POINTER(_vp), _u64, _vp]),
    "gl_free": (GlError, [_vp]),
    "gl_malloc_host": (GlError, [ctypes.POINTER(_vp), _u64]),
    "gl_free_host": (GlError, [_vp]),
    "gl_memcpy_h2d": (GlError, [_vp, _vp, _u64, _vp]),
    "gl_memcpy_h2d_async": (GlError, [_vp, _vp, _u64, _vp]),
    "gl_debug_copy": (GlError, [_vp, _vp, _u64, _vp]),
    "gl_memcpy_d2h": (GlError, [_vp, _vp, _u64, _vp]),
    "gl_memcpy_d2d": (GlError, [_vp, _vp, _u64, _vp]),
    "gl_memset_zero": (GlError, [_vp, _u64, _vp]),
    "gl_event_create": (GlError, [ctypes.POINTER(_vp)]),
    "gl_event_record": (GlError, [_vp, _vp]),
    "gl_event_elapsed_ms": (GlError, [ctypes.POINTER(ctypes.c_float), _vp, _vp]),
    "gl_event_destroy": (None, [_vp]),
    "gl_ntt_batch": (GlError, [_vp, _u64, _u32, _u64, _i, _i, _vp]),
    "gl_coset_lde_batch": (GlError, [_vp, _vp, _u64, _u32, _u32, _u64, _u64, _u64, _vp]),
    "gl_coset_ntt_batch": (GlError, [_vp, _u64, _u32, _u64, _u64, _i, _vp]),
    "gl_permutation_partial_products": (GlError, [_vp, _u64, _vp, _u64, _vp, _vp, _vp, _u32, _u32, _u32, _u32, _vp, _vp]),
    "gl_gate_programs_emit": (GlError, [_vp, _u32, _vp, _u32, _vp]),
    "gl_gate_programs_free": (None, [_vp]),
    "gl_gate_kernel_build": (GlError, [_vp, _u32, _vp, _u32, _vp, _u32, _u32, _u32, _u32, ctypes.POINTER(_vp)]),
    "gl_gate_kernel_destroy": (None, [_vp]),
    "gl_gate_kernel_source": (ctypes.c_char_p, [_vp]),
    "gl_circuit_create": (GlError, [ctypes.POINTER(GlCircuitDesc), ctypes.POINTER(_vp), _vp]),
    "gl_circuit_destroy": (None, [_vp]),
    "gl_circuit_trim": (GlError, [_vp]),
    "gl_circuit_info": (GlError, [_vp, _vp, _vp]),
    "gl_prove": (GlError, [_vp, _vp, _vp, _u32, ctypes.POINTER(_vp), ctypes.POINTER(_u64), _vp, _vp]),
    "gl_prove_many": (GlError, [_vp, _vp, _vp, _u32, _u32, _vp, _vp, _vp, _u32]),
    "gl_prove_zk": (GlError, [_vp, _vp, _vp, _u32, _vp, ctypes.POINTER(_vp), ctypes.POINTER(_u64), _vp, _vp]),
    "gl_bytes_free": (None, [_vp]),
    "gl_compute_quotient_polys": (GlError, [ctypes.POINTER(GlQuotientArgs), _vp, _vp]),
    "gl_eval_polys_ext2": (GlError, [_vp, _u64, _u32, _u64, _vp, _u32, _vp, _vp]),
    "gl_fri_reduce_polys_base": (GlError, [_vp, _u32, _u64, _vp, _vp, _vp]),
    "gl_fri_divide_by_linear": (GlError, [_vp, _u64, _vp, _vp, _i, _vp, _vp]),
    "gl_fri_fold": (GlError, [_vp, _u64, _u32, _vp, _vp, _vp]),
    "gl_ext2_interleave": (GlError, [_vp, _u64, _vp, _vp]),
    "gl_fri_proof_of_work": (GlError, [_vp, _u32, _u32, _vp, _vp]),
    "gl_poseidon_permute_batch": (GlError, [_vp, _u64, _vp]),
    "gl_sponge_absorb": (GlError, [_vp, _vp, _u32, _vp]),
    "gl_challenger_step": (GlError, [_vp, _vp, _u32, _u32, _vp, _u32, _vp]),
    "gl_fri_fold_device": (GlError, [_vp, _u64, _u32, _vp, _vp, _vp]),
    "gl_fri_proof_of_work_device": (GlError, [_vp, _u32, _vp, _vp, _vp]),
    "gl_merkle_open_batch_device": (GlError, [_vp, _u64, _u64, _u32, _u64, _u32, _vp, _vp, _u32, _u32, _vp, _vp, _vp]),
    "gl_merkle_open_batch": (GlError, [_vp, _u64, _u64, _u32, _u64, _u32, _vp, _vp, _u32, _vp, _vp, _vp]),
    "gl_merkle_tree_from_columns": (GlError, [_vp, _u32, _u64, _u64, _u32, _vp, _vp, _vp]),
    "gl_merkle_tree_from_leaves": (GlError, [_vp, _u32, _u64, _u32, _vp, _vp, _vp]),
    "gl_transpose": (GlError, [_vp, _vp, _u32, _u64, _u64, _vp]),
    "gl_commit_from_coeffs": (GlError, [_vp, _u64, _u32, _u32, _u32, _u32, _u64, _vp, _vp, _vp, _vp, _vp]),
    "gl_debug_field_op": (GlError, [_i, _vp, _vp, _vp, _u64, _vp]),
    "gl_commit_from_values": (GlError, [_vp, _u64, _u32, _u32, _u32, _u32, _u64, _vp, _vp, _vp, _vp, _vp]),
    # the reference's extern "C" surface (cuda/src/lib.rs:58-145)
    "init": (None, []),
    "ifft": (GlError, [_vp, _i, _i, _i, _vp, _vp, _vp]),
    "merkle_tree_from_coeffs": (GlError, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "merkle_tree_from_values": (GlError, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "build_merkle_tree": (GlError, [_vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "compute_quotient_polys": (GlError, [_vp, _i, _i, _i, _vp, _vp, _i, _i] + [_vp] * 12),
    "gl_reference_quotient_prepare": (GlError, [_vp]),
    "gl_reference_quotient_release": (GlError, []),
    "gl_reference_set_public_inputs_hash": (GlError, [_vp]),
    "gl_reference_set_public_inputs_hash_ctx": (GlError, [_vp, _vp]),
    "gl_reference_quotient_staging_bytes": (_u64, [_i]),
    "gl_reference_quotient_set_staging": (GlError, [_vp, _u64]),
    "cudaGetErrorString": (ctypes.c_char_p, [_i]),
}

_lib = None
_libc = ctypes.CDLL(None)
_libc.free.argtypes = [ctypes.c_void_p]


def load():
    """Load the shared library (no GPU needed for loading or symbol lookup)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(make -C plonky2_gpu_amd/csrc). There is no CPU fallback."
            )
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            f = getattr(lib, name)
            f.restype = res
            f.argtypes = args
        _lib = lib
    return _lib


def check(err):
    """Raise if a returned GlError carries a non-zero code; frees the message like the Rust Drop."""
    if err.code != 0:
        msg = ctypes.string_at(err.message).decode() if err.message else load().cudaGetErrorString(err.code).decode()
        if err.message:
            _libc.free(err.message)
        raise Plonky2HipError(err.code, msg)


def call(name, *args):
    """numpy arrays may be passed directly for host-pointer arguments (they stay alive for the call)."""
    import numpy as np

    conv = [a.ctypes.data if isinstance(a, np.ndarray) else a for a in args]
    check(getattr(load(), name)(*conv))
