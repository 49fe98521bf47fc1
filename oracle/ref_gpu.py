"""TEST INFRASTRUCTURE ONLY. ctypes loader of oracle/_ref/libplonky2_ref.so: the reference's own device kernels
(cuda/plonky2_gpu_impl.cuh, compiled unmodified for gfx950 by oracle/Makefile's `ref` target) behind oracle/ref_harness.hip.

Only tests/ and bench.py's reference leg may import this; nothing under plonky2_gpu_amd/ does (tests/test_abi.py enforces it).
All pointers are raw device addresses (ints) in the calling process; the harness allocates nothing."""
import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
PATH = os.path.join(HERE, "_ref", "libplonky2_ref.so")
REFERENCE_CUDA = "/root/reference/cuda"
P = 0xFFFFFFFF00000001

_lib = None

_u64p = ctypes.c_void_p
_i = ctypes.c_int
_u64 = ctypes.c_uint64
_fp = ctypes.POINTER(ctypes.c_float)

SIGNATURES = {
    "ref_ifft": [_u64p, _i, _i, _i, _u64p, _u64, _fp],
    "ref_fft": [_u64p, _i, _i, _i, _u64p, _i, _fp],
    "ref_coset_lde": [_u64p, _u64p, _i, _i, _i, _u64p, _u64p, _i, _fp],
    "ref_reverse_index_bits": [_u64p, _i, _i, _i, _fp],
    "ref_merkle_tree": [_u64p, _i, _i, _i, _fp],
    "ref_transpose": [_u64p, _u64p, _i, _i, _fp],
    "ref_compute_quotient_polys": [_u64p, _i, _i] + [_u64p] * 13 + [ctypes.POINTER(_u64), _u64, _fp],
    "ref_compute_quotient_values": [_u64p, _i, _i] + [_u64p] * 10 + [ctypes.POINTER(_u64), _fp],
}


class ReferenceKernelError(RuntimeError):
    pass


def available():
    return os.path.exists(PATH)


def why_absent():
    return ("oracle/_ref/libplonky2_ref.so is absent: it is built by `make -C oracle ref` (called from "
            "__graft_entry__.build()) only where the reference's sources are mounted at %s" % REFERENCE_CUDA)


def build():
    """(re)build when the reference is mounted; returns True when the library exists afterwards"""
    import subprocess

    if os.path.isdir(REFERENCE_CUDA):
        subprocess.check_call(["make", "-C", HERE, "ref"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return available()


def lib():
    global _lib
    if _lib is None:
        if not available():
            raise ReferenceKernelError(why_absent())
        _lib = ctypes.CDLL(PATH)
        for name, args in SIGNATURES.items():
            f = getattr(_lib, name)
            f.argtypes = args
            f.restype = ctypes.c_int
        _lib.ref_error_string.restype = ctypes.c_char_p
        _lib.ref_error_string.argtypes = [ctypes.c_int]
    return _lib


def call(name, *args, n_ms=1):
    """-> list of kernel durations in ms (HIP events inside the harness)"""
    ms = (ctypes.c_float * n_ms)()
    rc = getattr(lib(), name)(*args, ms)
    if rc:
        raise ReferenceKernelError("%s: HIP error %d (%s)" % (name, rc, lib().ref_error_string(rc).decode()))
    return list(ms)


def n_inv(log_n):
    """1/2^log_n mod p, the value the reference's host passes (cuda/plonky2_gpu.cu:77, :746)"""
    return P - ((P - 1) >> log_n)
