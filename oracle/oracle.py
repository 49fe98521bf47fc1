"""ctypes binding of the C oracle (oracle/libgl_oracle.so) — TEST INFRASTRUCTURE ONLY.

Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never from the
product package. Arrays are numpy uint64.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libgl_oracle.so")

P = 0xFFFFFFFF00000001


def build(force=False):
    src = [os.path.join(_HERE, f) for f in ("gl_oracle.c", "gl_oracle.h", "prove_oracle.c", "prove_oracle.h", "poseidon_constants.h")]
    if (
        force
        or not os.path.exists(_LIB_PATH)
        or any(os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in src)
    ):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libgl_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None
_u64p = ctypes.POINTER(ctypes.c_uint64)


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        u64, sz, ui, i = ctypes.c_uint64, ctypes.c_size_t, ctypes.c_uint, ctypes.c_int
        for name, res, args in [
            ("glo_add", u64, [u64, u64]),
            ("glo_sub", u64, [u64, u64]),
            ("glo_neg", u64, [u64]),
            ("glo_mul", u64, [u64, u64]),
            ("glo_canon", u64, [u64]),
            ("glo_mac", u64, [u64, u64, u64]),
            ("glo_exp", u64, [u64, u64]),
            ("glo_inverse", u64, [u64]),
            ("glo_inverse_2exp", u64, [ui]),
            ("glo_primitive_root_of_unity", u64, [ui]),
            ("glo_reverse_bits", sz, [sz, ui]),
            ("glo_reverse_index_bits_in_place", None, [_u64p, sz]),
            ("glo_reverse_index_bits_rows_in_place", None, [_u64p, sz, sz]),
            ("glo_transpose", None, [_u64p, _u64p, sz, sz]),
            ("glo_fft_root_table_concat", sz, [sz, _u64p]),
            ("glo_fft", None, [_u64p, sz, ui]),
            ("glo_ifft", None, [_u64p, sz]),
            ("glo_coset_lde", None, [_u64p, sz, ui, u64, _u64p]),
            ("glo_coset_fft", None, [_u64p, sz, u64]),
            ("glo_coset_ifft", None, [_u64p, sz, u64]),
            ("glo_poseidon", None, [_u64p]),
            ("glo_poseidon_naive", None, [_u64p]),
            ("glo_hash_no_pad", None, [_u64p, sz, _u64p]),
            ("glo_hash_or_noop", None, [_u64p, sz, _u64p]),
            ("glo_two_to_one", None, [_u64p, _u64p, _u64p]),
            ("glo_merkle_tree", i, [_u64p, sz, sz, ui, _u64p, _u64p, i]),
            ("glo_merkle_prove", ui, [_u64p, sz, ui, sz, _u64p]),
            ("glo_merkle_verify", i, [_u64p, sz, sz, _u64p, _u64p, ui]),
            ("glo_commit_from_values", i, [_u64p, sz, sz, ui, ui, _u64p, _u64p, _u64p, _u64p, i]),
            ("glo_commit_from_coeffs", i, [_u64p, sz, sz, ui, ui, _u64p, _u64p, _u64p, i]),
            ("glo_fft_batch", None, [_u64p, sz, sz, i, i]),
            ("glo_coset_lde_batch", None, [_u64p, sz, sz, ui, u64, _u64p, i]),
            ("glo_fft_bench", ctypes.c_double, [sz, i, i, ctypes.c_uint64, _u64p]),
            ("glo_hardware_threads", i, []),
        ]:
            f = getattr(L, name)
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(_u64p) if a is not None else None


def _arr(x):
    return np.ascontiguousarray(np.asarray(x, dtype=np.uint64))


def canon(a):
    a = _arr(a)
    return np.where(a >= np.uint64(P), a - np.uint64(P), a)


def fft(v, r=0):
    out = _arr(v).copy()
    lib().glo_fft(_p(out), out.size, r)
    return out


def ifft(v):
    out = _arr(v).copy()
    lib().glo_ifft(_p(out), out.size)
    return out


def coset_lde(coeffs, rate_bits, shift=7):
    c = _arr(coeffs)
    out = np.empty(c.size << rate_bits, dtype=np.uint64)
    lib().glo_coset_lde(_p(c), c.size, rate_bits, shift, _p(out))
    return out


def coset_lde_batch(coeffs, rate_bits, shift=7, threads=1):
    """coeffs [n_polys, n] -> [n_polys, n << rate_bits] (natural order), one column per thread."""
    c = _arr(coeffs)
    out = np.empty((c.shape[0], c.shape[1] << rate_bits), dtype=np.uint64)
    lib().glo_coset_lde_batch(_p(c), c.shape[0], c.shape[1], rate_bits, shift, _p(out), threads)
    return out


def coset_fft(v, shift=7):
    out = _arr(v).copy()
    lib().glo_coset_fft(_p(out), out.size, shift)
    return out


def coset_ifft(v, shift=7):
    out = _arr(v).copy()
    lib().glo_coset_ifft(_p(out), out.size, shift)
    return out


def fft_batch(v, inverse=False, threads=1):
    """v: [n_polys, n] (row = one column polynomial)."""
    out = _arr(v).copy()
    lib().glo_fft_batch(_p(out), out.shape[0], out.shape[1], int(inverse), threads)
    return out


def fft_bench(n, threads, cols_per_thread=1, seed=0x706C6F6E6B7932):
    """Seconds for threads * cols_per_thread forward + inverse transforms of length n (2x that many NTTs), clocked
    inside C: root table prebuilt, every thread working on columns it allocated and filled itself. Raises if any
    column failed to come back unchanged."""
    bad = np.zeros(1, dtype=np.uint64)
    dt = lib().glo_fft_bench(n, threads, cols_per_thread, seed, _p(bad))
    if int(bad[0]):
        raise AssertionError("ifft(fft(x)) != x in the CPU restatement")
    return dt


def root_table_concat(n):
    k = lib().glo_fft_root_table_concat(n, None)
    out = np.empty(k, dtype=np.uint64)
    lib().glo_fft_root_table_concat(n, _p(out))
    return out


def poseidon(state, naive=False):
    s = _arr(state).copy()
    assert s.size == 12
    (lib().glo_poseidon_naive if naive else lib().glo_poseidon)(_p(s))
    return s


def hash_or_noop(x):
    x = _arr(x)
    out = np.empty(4, dtype=np.uint64)
    lib().glo_hash_or_noop(_p(x), x.size, _p(out))
    return out


def hash_no_pad(x):
    x = _arr(x)
    out = np.empty(4, dtype=np.uint64)
    lib().glo_hash_no_pad(_p(x), x.size, _p(out))
    return out


def two_to_one(l, r):
    l, r = _arr(l), _arr(r)
    out = np.empty(4, dtype=np.uint64)
    lib().glo_two_to_one(_p(l), _p(r), _p(out))
    return out


def merkle_tree(leaves, cap_height, threads=1):
    """leaves: [n_leaves, leaf_len]. Returns (digests [num_digests,4], cap [2^h,4])."""
    lv = _arr(leaves)
    n, ll = lv.shape
    if cap_height > n.bit_length() - 1:
        raise ValueError("cap_height should be at most log2(leaves.len())")
    nd = 2 * (n - (1 << cap_height))
    dig = np.empty((nd, 4), dtype=np.uint64)
    cap = np.empty((1 << cap_height, 4), dtype=np.uint64)
    rc = lib().glo_merkle_tree(_p(lv), n, ll, cap_height, _p(dig), _p(cap), threads)
    assert rc == 0
    return dig, cap


def merkle_prove(digests, n_leaves, cap_height, leaf_index):
    dg = _arr(digests)
    sib = np.empty((64, 4), dtype=np.uint64)
    k = lib().glo_merkle_prove(_p(dg), n_leaves, cap_height, leaf_index, _p(sib))
    return sib[:k].copy()


def merkle_verify(leaf, leaf_index, cap, siblings):
    leaf, cap, sib = _arr(leaf), _arr(cap), _arr(siblings)
    return bool(lib().glo_merkle_verify(_p(leaf), leaf.size, leaf_index, _p(cap), _p(sib), sib.shape[0] if sib.ndim == 2 else 0))


def commit_from_values(values, rate_bits, cap_height, threads=1, want_leaves=True):
    """values [n_polys, n]. Returns dict(coeffs, leaves, digests, cap)."""
    v = _arr(values)
    P_, n = v.shape
    n_ext = n << rate_bits
    coeffs = np.empty_like(v)
    leaves = np.empty((n_ext, P_), dtype=np.uint64) if want_leaves else None
    nd = 2 * (n_ext - (1 << cap_height))
    dig = np.empty((nd, 4), dtype=np.uint64)
    cap = np.empty((1 << cap_height, 4), dtype=np.uint64)
    rc = lib().glo_commit_from_values(_p(v), P_, n, rate_bits, cap_height, _p(coeffs), _p(leaves), _p(dig), _p(cap), threads)
    if rc != 0:
        raise ValueError("commit failed rc=%d" % rc)
    return dict(coeffs=coeffs, leaves=leaves, digests=dig, cap=cap)


def commit_from_coeffs(coeffs, rate_bits, cap_height, threads=1, want_leaves=True):
    c = _arr(coeffs)
    P_, n = c.shape
    n_ext = n << rate_bits
    leaves = np.empty((n_ext, P_), dtype=np.uint64) if want_leaves else None
    nd = 2 * (n_ext - (1 << cap_height))
    dig = np.empty((nd, 4), dtype=np.uint64)
    cap = np.empty((1 << cap_height, 4), dtype=np.uint64)
    rc = lib().glo_commit_from_coeffs(_p(c), P_, n, rate_bits, cap_height, _p(leaves), _p(dig), _p(cap), threads)
    if rc != 0:
        raise ValueError("commit failed rc=%d" % rc)
    return dict(leaves=leaves, digests=dig, cap=cap)


def hardware_threads():
    return lib().glo_hardware_threads()


def usable_threads():
    """threads worth starting: the hardware threads this process may use, capped by the container's CPU quota"""
    q = cpu_quota()
    return max(1, min(hardware_threads(), int(q))) if q else hardware_threads()


def cpu_quota():
    """CPUs' worth of time the container may use, from the cgroup (v2 cpu.max, v1 cpu.cfs_quota_us / cpu.cfs_period_us);
    None when there is no limit. The GPU boxes of the pool grant 16 of the host's 256 hardware threads."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / per
    except (OSError, ValueError):
        return None


def random_field(shape, seed=0x706C6F6E6B7932):
    """Uniform canonical field elements from a seeded generator (numpy PCG64; rejection-free:
    draw 64 bits and fold the 2^32-1 values >= p back by subtraction — bias 2^-32, irrelevant
    for tests; the SplitMix64 stream of SURVEY §8d is in pyref.splitmix64 for the fixtures)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    a = rng.integers(0, 2**64, size=shape, dtype=np.uint64)
    return np.where(a >= np.uint64(P), a - np.uint64(P), a)
