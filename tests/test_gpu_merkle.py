"""HIP Poseidon / Merkle tree / PolynomialBatch commit vs the CPU oracle and the reference's
known answers. Bit-exact."""
import ctypes

import numpy as np
import pytest

from gpu_util import P, bitrev_perm, gpu  # noqa: F401
from test_oracle_poseidon import TEST_VECTORS

pytestmark = pytest.mark.gpu


def permute(gpu, states):
    import plonky2_gpu_amd as pg
    from plonky2_gpu_amd import _lib

    s = np.ascontiguousarray(states, dtype=np.uint64).reshape(-1, 12)
    buf = pg.DeviceBuffer.from_host(gpu, s)
    _lib.call("gl_poseidon_permute_batch", buf.ptr, s.shape[0], gpu.ptr)
    return buf.download().reshape(-1, 12)


def test_poseidon_known_answers(gpu):
    # plonky2/src/hash/poseidon_goldilocks.rs:286-309
    inp = np.array([v[0] for v in TEST_VECTORS], dtype=np.uint64)
    exp = np.array([v[1] for v in TEST_VECTORS], dtype=np.uint64)
    assert (permute(gpu, inp) == exp).all()


def test_poseidon_random_and_noncanonical(gpu, oracle):
    x = oracle.random_field((1000, 12), seed=11)
    x[0, :] = np.uint64(2**64 - 1)
    x[1, :] = np.uint64(P)
    x[2, ::2] = np.uint64(P + 5)
    got = permute(gpu, x)
    for i in range(0, 1000, 7):
        assert (got[i] == oracle.canon(oracle.poseidon(x[i]))).all(), i
    for i in range(3):
        assert (got[i] == oracle.canon(oracle.poseidon(x[i]))).all(), i


def test_poseidon_edge_states_and_ragged_counts(gpu, oracle):
    """The matrix-core MDS layer (csrc/poseidon.h) works on the state's BYTES (as byte - 128) and on whole waves: states made of
    the bytes where that could go wrong (0x00, 0x7f, 0x80, 0xff, mixed), counts that leave a wave partly or almost empty."""
    pats = [0x0000000000000000, 0xFFFFFFFFFFFFFFFF, 0x8080808080808080, 0x7F7F7F7F7F7F7F7F, 0xFF00FF00FF00FF00, 0x00FF00FF00FF00FF,
            0x80FF7F0001FE8081, 0xFFFFFFFF00000000, 0xFFFFFFFF00000001, 0x00000000FFFFFFFF]
    x = np.array([[pats[(i + 3 * j) % len(pats)] for j in range(12)] for i in range(2 * len(pats))], dtype=np.uint64)
    x[len(pats):] = np.array(pats, dtype=np.uint64)[:, None]  # all twelve words the same
    got = permute(gpu, x)
    for i in range(x.shape[0]):
        assert (got[i] == oracle.canon(oracle.poseidon(x[i]))).all(), i
    rnd = oracle.random_field((321, 12), seed=5)
    exp = np.array([oracle.canon(oracle.poseidon(r)) for r in rnd[:70]])
    for count in (1, 2, 63, 65, 129, 321):
        got = permute(gpu, rnd[:count])
        assert (got[:min(count, 70)] == exp[:min(count, 70)]).all(), count
        assert (got[-1] == oracle.canon(oracle.poseidon(rnd[count - 1]))).all(), count


def test_two_poseidon_implementations_agree(gpu):
    """gl_poseidon_permute_batch from the product (all thirty MDS layers on the matrix cores, `poseidon_naive`'s round structure)
    and from the diagnostic build under PLONKY2_POSEIDON=vector (vector ALU, the "fast" partial rounds in blocks of eleven:
    csrc/poseidon_vector.h) on 2^16 random 64-bit states, canonical or not — bit for bit."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = f"""
import sys, numpy as np, hashlib
sys.path.insert(0, {root!r})
import plonky2_gpu_amd as pg
from plonky2_gpu_amd import _lib
ctx = pg.Context(0)
x = np.random.default_rng(17).integers(0, 2**64, size=(1 << 16, 12), dtype=np.uint64)
buf = pg.DeviceBuffer.from_host(ctx, x)
_lib.call("gl_poseidon_permute_batch", buf.ptr, x.shape[0], ctx.ptr)
print("digest", hashlib.sha256(buf.download().tobytes()).hexdigest())
"""
    outs = []
    for env in ({}, {"PLONKY2_HIP_LIBRARY": os.path.join(root, "plonky2_gpu_amd", "libplonky2_hip_debug.so"), "PLONKY2_POSEIDON": "vector"}):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-2000:]
        outs.append(r.stdout.strip().splitlines()[-1])
    assert outs[0] == outs[1] and outs[0].startswith("digest ")


@pytest.mark.parametrize("n,k,h", [(256, 7, 1), (256, 7, 8), (256, 7, 0), (2, 5, 1), (1, 9, 0), (16, 4, 2), (16, 3, 0),
                                   (8, 1, 1), (64, 8, 3), (64, 9, 3), (32, 16, 2), (128, 135, 4), (4096, 20, 4),
                                   (1024, 88, 10), (512, 17, 5)])
def test_merkle_tree_matches_oracle(gpu, oracle, n, k, h):
    import plonky2_gpu_amd as pg
    from plonky2_gpu_amd import _lib

    leaves = oracle.random_field((n, k), seed=n * 131 + k * 7 + h)
    leaves[0, 0] = np.uint64(2**64 - 1)  # non-canonical input
    dig, cap = oracle.merkle_tree(leaves, h, threads=4)
    dig, cap = oracle.canon(dig), oracle.canon(cap)
    tree = pg.MerkleTree.new(gpu, leaves, h)
    assert (tree.cap == cap).all()
    assert tree.digests.shape == dig.shape and (tree.digests == dig).all()
    # the same tree from the column-major layout the NTT produces
    cols = pg.DeviceBuffer.from_host(gpu, np.ascontiguousarray(leaves.T))
    d2 = pg.DeviceBuffer(gpu, max(dig.size, 1))
    c2 = pg.DeviceBuffer(gpu, cap.size)
    _lib.call("gl_merkle_tree_from_columns", cols.ptr, k, n, n, h, d2.ptr, c2.ptr, gpu.ptr)
    assert (c2.download().reshape(-1, 4) == cap).all()
    assert (d2.download(0, dig.size).reshape(-1, 4) == dig).all()
    # merkle_tree.rs:456-468: every (sampled) leaf's proof verifies against the cap
    for i in sorted(set([0, 1, n // 2, n - 1] + list(range(0, n, max(1, n // 16))))):
        if i < n:
            assert oracle.merkle_verify(leaves[i], i, tree.cap, tree.prove(i))


def test_merkle_cap_height_too_big(gpu, oracle):
    # merkle_tree.rs:470-482 (should_panic) -> ValueError in the mirror, GL_E_INVALID at the C ABI
    import plonky2_gpu_amd as pg
    from plonky2_gpu_amd import _lib

    leaves = oracle.random_field((256, 7))
    with pytest.raises(ValueError):
        pg.MerkleTree.new(gpu, leaves, 9)
    buf = pg.DeviceBuffer.from_host(gpu, leaves)
    with pytest.raises(pg.Plonky2HipError) as e:
        _lib.call("gl_merkle_tree_from_leaves", buf.ptr, 7, 256, 9, buf.ptr, buf.ptr, gpu.ptr)
    assert e.value.code == pg.GL_E_INVALID


@pytest.mark.parametrize("n_polys,log_n,rate_bits,h", [(5, 4, 3, 2), (135, 6, 3, 4), (3, 0, 3, 1), (20, 10, 3, 4), (2, 13, 3, 4),
                                                       (16, 14, 1, 0), (234, 8, 3, 4), (4, 5, 3, 8), (9, 12, 2, 4)])
def test_commit_from_values_matches_oracle(gpu, oracle, n_polys, log_n, rate_bits, h):
    """PolynomialBatch::from_values (fri/oracle.rs:709-731): coefficients, leaf-major leaves,
    digests and cap all equal the CPU path's."""
    import plonky2_gpu_amd as pg

    vals = oracle.random_field((n_polys, 1 << log_n), seed=n_polys * 1000 + log_n * 10 + rate_bits)
    exp = oracle.commit_from_values(vals, rate_bits, h, threads=4)
    batch = pg.PolynomialBatch.from_values(gpu, vals, rate_bits, False, h)
    assert (batch.polynomials == oracle.canon(exp["coeffs"])).all()
    assert (batch.merkle_tree.cap == oracle.canon(exp["cap"])).all()
    assert (batch.merkle_tree.digests == oracle.canon(exp["digests"])).all()
    leaves = batch.merkle_tree.d_leaves.download().reshape(-1, n_polys)
    assert (leaves == oracle.canon(exp["leaves"])).all()
    assert (batch.lde_column_major() == oracle.canon(exp["leaves"]).T).all()
    # get_lde_values (oracle.rs:1007-1018) returns the row of the natural-order point
    n_ext = 1 << (log_n + rate_bits)
    for idx in (0, 1, n_ext - 1, n_ext // 3):
        nat = oracle.canon(oracle.coset_lde(oracle.canon(exp["coeffs"])[0], rate_bits))[idx]
        assert batch.get_lde_values(idx)[0] == nat
    # from_coeffs on the same coefficients gives the same commitment (oracle.rs:911-977)
    b2 = pg.PolynomialBatch.from_coeffs(gpu, batch.polynomials, rate_bits, False, h, leaf_major=False)
    assert (b2.merkle_tree.cap == batch.merkle_tree.cap).all()


@pytest.mark.parametrize(
    "n_polys,log_n,rate_bits,h,leaf_major,salted",
    [
        (48, 13, 3, 4, True, False),  # exactly three chunks, whole rate blocks
        (50, 13, 3, 4, False, False),  # last chunk of 2 columns: merged with the one before it
        (55, 14, 2, 0, True, False),  # 7 columns left over, cap height 0
        (57, 13, 3, 2, False, False),  # 9 left over: its own launch, ragged last block
        (135, 13, 3, 4, True, False),  # the benchmark's leaf length
        (64, 15, 1, 4, False, False),
        (50, 13, 3, 4, True, True),  # salt columns ride on the last launch
        (60, 13, 3, 3, False, True),  # 60 + 4 = whole rate blocks only with the salt
    ],
)
def test_commit_through_the_pipelined_path(gpu, oracle, n_polys, log_n, rate_bits, h, leaf_major, salted):
    """From 48 columns and 2^16 leaves on, the commit runs the LDE in chunks of 16 columns on the caller's stream while a second
    stream absorbs each finished chunk into the leaves' sponges (capi.hip commit_from_coeffs_impl; the capacity words wait in the
    digest slot between launches). Same answers as the one-shot path's oracle: coefficients, LDE, digests, cap, leaf-major copy,
    for chunk counts and leftovers of every kind, with and without salt columns."""
    import plonky2_gpu_amd as pg

    n_ext = 1 << (log_n + rate_bits)
    assert n_polys >= 48 and n_ext >= 1 << 16  # the conditions of the pipelined branch
    vals = oracle.random_field((n_polys, 1 << log_n), seed=n_polys * 77 + log_n)
    salt = oracle.random_field((4, n_ext), seed=n_polys + 1) if salted else None
    batch = pg.PolynomialBatch.from_values(gpu, vals, rate_bits, salted, h, salt=salt, leaf_major=leaf_major)
    exp = oracle.commit_from_values(vals, rate_bits, h, threads=8)
    leaves = oracle.canon(exp["leaves"])
    if salted:
        leaves = np.concatenate([leaves, salt.T], axis=1)
        dig, cap = oracle.merkle_tree(leaves, h, threads=8)
    else:
        dig, cap = exp["digests"], exp["cap"]
    assert (batch.polynomials == oracle.canon(exp["coeffs"])).all()
    assert (batch.merkle_tree.cap == oracle.canon(cap)).all()
    assert (batch.merkle_tree.digests == oracle.canon(dig).reshape(-1, 4)).all()
    assert (batch.lde_column_major()[:n_polys] == leaves[:, :n_polys].T).all()
    if leaf_major:
        assert (batch.merkle_tree.d_leaves.download().reshape(n_ext, -1) == leaves).all()
    # from_coeffs takes the same branch
    b2 = pg.PolynomialBatch.from_coeffs(gpu, batch.polynomials, rate_bits, salted, h, salt=salt, leaf_major=False)
    assert (b2.merkle_tree.cap == batch.merkle_tree.cap).all() and (b2.merkle_tree.digests == batch.merkle_tree.digests).all()


def test_commit_with_blinding_salt(gpu, oracle):
    """blinding appends SALT_SIZE columns to every leaf (oracle.rs:985-1002); with caller-provided
    salt the tree equals the oracle's tree over [LDE | salt]."""
    import plonky2_gpu_amd as pg

    n_polys, log_n, rate_bits, h = 6, 5, 3, 2
    n_ext = 1 << (log_n + rate_bits)
    vals = oracle.random_field((n_polys, 1 << log_n), seed=9)
    salt = oracle.random_field((4, n_ext), seed=10)
    batch = pg.PolynomialBatch.from_values(gpu, vals, rate_bits, True, h, salt=salt)
    exp = oracle.commit_from_values(vals, rate_bits, h)
    leaves = np.concatenate([oracle.canon(exp["leaves"]), salt.T], axis=1)
    dig, cap = oracle.merkle_tree(leaves, h)
    assert (batch.merkle_tree.cap == oracle.canon(cap)).all()
    assert (batch.merkle_tree.digests == oracle.canon(dig)).all()
    assert (batch.merkle_tree.d_leaves.download().reshape(n_ext, -1) == leaves).all()
    assert len(batch.get_lde_values(3)) == n_polys


def test_reference_abi_entry_points(gpu, oracle):
    """The drop-in symbols with the reference's region contract (cuda/plonky2_gpu.cu:435-606):
    ext[0..] leaf-major, ext[pad..] column-major bit-reversed LDE, then digests || cap."""
    import plonky2_gpu_amd as pg
    from plonky2_gpu_amd import _lib

    P_, log_n, rate_bits, h = 20, 9, 3, 4
    n, n_ext = 1 << log_n, 1 << (log_n + rate_bits)
    vals = oracle.random_field((P_, n), seed=31)
    exp = oracle.commit_from_values(vals, rate_bits, h, threads=4)
    pad = P_ * n_ext
    nd = 2 * (n_ext - (1 << h))
    total = 2 * pad + 4 * nd + 4 * (1 << h)
    ext = pg.DeviceBuffer(gpu, total)
    ext.upload(vals, 0)  # values live at the start of the region, as in oracle.rs:352-362
    n_inv = ctypes.c_uint64(P - ((P - 1) >> log_n))
    _lib.call("ifft", ext.ptr, P_, n, log_n, None, ctypes.addressof(n_inv), gpu.ptr)
    assert (ext.download(0, P_ * n).reshape(P_, n) == oracle.canon(exp["coeffs"])).all()
    _lib.call("merkle_tree_from_coeffs", ext.ptr, ext.ptr, P_, n, log_n, None, None, None, rate_bits, 0, h, pad, gpu.ptr)
    leaves = ext.download(0, pad).reshape(n_ext, P_)
    assert (leaves == oracle.canon(exp["leaves"])).all()
    colmajor = ext.download(pad, pad).reshape(P_, n_ext)
    assert (colmajor == oracle.canon(exp["leaves"]).T).all()
    dig = ext.download(2 * pad, 4 * nd).reshape(-1, 4)
    cap = ext.download(2 * pad + 4 * nd, 4 << h).reshape(-1, 4)
    assert (dig == oracle.canon(exp["digests"])).all() and (cap == oracle.canon(exp["cap"])).all()
    # merkle_tree_from_values = ifft + merkle_tree_from_coeffs (the reference's body is assert(0))
    ext2 = pg.DeviceBuffer(gpu, total)
    ext2.upload(vals, 0)
    _lib.call("merkle_tree_from_values", ext2.ptr, ext2.ptr, P_, n, log_n, None, None, None, ctypes.addressof(n_inv),
              rate_bits, 0, h, pad, gpu.ptr)
    assert (ext2.download(2 * pad + 4 * nd, 4 << h).reshape(-1, 4) == cap).all()
    # build_merkle_tree: natural-order LDE resident at ext+pad -> same tree
    ext3 = pg.DeviceBuffer(gpu, total)
    nat = np.stack([oracle.canon(oracle.coset_lde(oracle.canon(exp["coeffs"])[i], rate_bits)) for i in range(P_)])
    ext3.upload(nat, pad)
    _lib.call("build_merkle_tree", ext3.ptr, P_, n, log_n, rate_bits, 0, h, pad, gpu.ptr)
    assert (ext3.download(2 * pad + 4 * nd, 4 << h).reshape(-1, 4) == cap).all()
    # wrong n_inv is rejected instead of silently ignored; the quotient entry (tests/test_reference_quotient.py)
    # rejects null arguments
    bad = ctypes.c_uint64(12345)
    with pytest.raises(pg.Plonky2HipError):
        _lib.call("ifft", ext.ptr, P_, n, log_n, None, ctypes.addressof(bad), gpu.ptr)
    with pytest.raises(pg.Plonky2HipError) as e:
        _lib.call("compute_quotient_polys", None, 0, 0, 0, None, None, 0, 0, *([None] * 12))
    assert e.value.code == pg.GL_E_INVALID


def test_open_batch_equals_get_and_prove(gpu, oracle):
    import plonky2_gpu_amd as pg

    rng = np.random.default_rng(5)
    for n, ll, cap_h in ((64, 7, 2), (256, 135, 0), (16, 3, 4), (32, 20, 1)):
        leaves = rng.integers(0, P, size=(n, ll), dtype=np.uint64)
        tree = pg.MerkleTree.new(gpu, leaves, cap_h)
        idx = [0, n - 1, 5 % n, 5 % n, n // 2]
        lv, sib = tree.open_batch(idx)
        for q, i in enumerate(idx):
            assert (lv[q] == leaves[i]).all()
            assert (sib[q] == tree.prove(i)).all()
            assert oracle.merkle_verify(leaves[i], i, tree.cap, sib[q])


def test_sponge_absorb_matches_hash_no_pad(gpu):
    from oracle import pyref
    from plonky2_gpu_amd.challenger import hash_no_pad

    for k in (1, 3, 8, 9, 16, 23, 135):
        inp = [(i * 0x9E3779B97F4A7C15 + 7) % P for i in range(k)]
        assert hash_no_pad(gpu, inp) == pyref.hash_no_pad(inp)


def test_open_batch_from_the_column_major_lde(gpu):
    """a commitment built without the leaf-major copy still opens leaves + paths (strided gather)"""
    import plonky2_gpu_amd as pg

    rng = np.random.default_rng(8)
    vals = rng.integers(0, P, size=(9, 64), dtype=np.uint64)
    a = pg.PolynomialBatch.from_values(gpu, vals, 3, False, 2, leaf_major=True)
    b = pg.PolynomialBatch.from_values(gpu, vals, 3, False, 2, leaf_major=False)
    assert b.merkle_tree.d_leaves is None and (a.merkle_tree.cap == b.merkle_tree.cap).all()
    idx = [0, 511, 37, 256]
    la, sa = a.merkle_tree.open_batch(idx)
    lb, sb = b.merkle_tree.open_batch(idx)
    assert (la == lb).all() and (sa == sb).all()
    assert (b.get_lde_values(5) == a.get_lde_values(5)).all()


def test_commit_at_the_full_benchmark_size(gpu, oracle):
    """BASELINE.json configs[2], the shape bench.py's commit leg times: from_values of 135 columns x 2^20 rows, rate 8,
    cap height 4 — LDE 2^23 x 135 = 8.4 GiB, 2^23 leaves of 17 permutations each. Checked through properties that do not
    need the whole answer on the host, AND against the whole answer (the oracle's cap and all its digests):
      - three whole columns (first, middle, last): coefficients == the C oracle's ifft of the values, and the LDE
        column == its 2^23-point coset LDE in bit-reversed order;
      - from_coeffs on the coefficients builds the same tree (same cap, same sampled digests);
      - 50 opened leaves and their paths verify against the cap with the oracle's Merkle verifier, and the opened
        leaves carry the LDE columns' values at those rows."""
    import plonky2_gpu_amd as pg
    from plonky2_gpu_amd import _lib

    n_polys, log_n, rate_bits, h = 135, 20, 3, 4
    n, n_ext = 1 << log_n, 1 << (log_n + rate_bits)
    watch = (0, 67, 134)
    d_vals = pg.DeviceBuffer(gpu, n_polys * n)
    values = {}
    all_values = np.empty((n_polys, n), dtype=np.uint64)
    for c0 in range(0, n_polys, 15):
        v = oracle.random_field((15, n), seed=4200 + c0)
        d_vals.upload(v, c0 * n)
        all_values[c0:c0 + 15] = v
        for c in watch:
            if c0 <= c < c0 + 15:
                values[c] = v[c - c0].copy()
    a = pg.PolynomialBatch.from_values_device(gpu, d_vals, n_polys, log_n, rate_bits, False, h, leaf_major=False)
    # THE WHOLE ANSWER: the C oracle's commit of the same 135 x 2^20 values on every core this process may use (151 M
    # permutations, 135 transforms of 2^23 points: some tens of seconds) -> its cap and every one of its 2 (2^23 - 16) digests
    import time

    quota = oracle.cpu_quota()
    threads = max(1, min(oracle.hardware_threads(), int(quota))) if quota else oracle.hardware_threads()
    t0 = time.perf_counter()
    exp = oracle.commit_from_values(all_values, rate_bits, h, threads=threads, want_leaves=False)
    print("oracle commit of configs[2] on %d threads: %.1f s = %.2f M leaves/s" % (threads, time.perf_counter() - t0, n_ext / (time.perf_counter() - t0) / 1e6))
    del all_values
    assert (a.merkle_tree.cap == oracle.canon(exp["cap"])).all(), "cap of configs[2] at full size"
    assert (a.merkle_tree.d_digests.download().reshape(-1, 4) == oracle.canon(exp["digests"])).all(), "digests of configs[2] at full size"
    del exp
    perm = bitrev_perm(log_n + rate_bits)
    lde = {}
    for c in watch:
        coeffs = a.d_polynomials.download(c * n, n)
        assert (coeffs == oracle.canon(oracle.ifft(values[c]))).all(), c
        lde[c] = a.d_lde.download(c * n_ext, n_ext)
        assert (lde[c] == oracle.canon(oracle.coset_lde(coeffs, rate_bits))[perm]).all(), c
    cap = a.merkle_tree.cap
    assert cap.shape == (16, 4)

    rng = np.random.default_rng(77)
    idx = [0, n_ext - 1] + [int(i) for i in rng.integers(0, n_ext, size=48)]
    leaves, sib = a.merkle_tree.open_batch(idx)
    for q, i in enumerate(idx):
        assert oracle.merkle_verify(leaves[q], i, cap, sib[q]), i
        for c in watch:
            assert leaves[q][c] == lde[c][i], (i, c)
    # a corrupted leaf must not verify (the verifier is not vacuous at this size)
    bad = leaves[0].copy()
    bad[5] ^= np.uint64(1)
    assert not oracle.merkle_verify(bad, idx[0], cap, sib[0])

    d_coeffs = pg.DeviceBuffer(gpu, n_polys * n)
    _lib.call("gl_memcpy_d2d", d_coeffs.ptr, a.d_polynomials.ptr, 8 * n_polys * n, gpu.ptr)
    b = pg.PolynomialBatch.from_coeffs_device(gpu, d_coeffs, n_polys, log_n, rate_bits, False, h, leaf_major=False)
    assert (b.merkle_tree.cap == cap).all()
    for slot in [0, 1, 2 * (n_ext - 16) - 1] + [int(s) for s in rng.integers(0, 2 * (n_ext - 16), size=200)]:
        assert (a.merkle_tree.d_digests.download(4 * slot, 4) == b.merkle_tree.d_digests.download(4 * slot, 4)).all(), slot


def _bitrev_take(col, log_n):
    """col[bitrev(i)] for all i, without a 64-bit index array per call site (2^26 entries at the largest size)"""
    perm = bitrev_perm(log_n)
    out = col[perm]
    del perm
    return out


@pytest.mark.parametrize("log_n", [21, 22, 23])
def test_full_width_commit_of_large_traces(gpu, oracle, log_n):
    """north_star's trace sizes at the FULL width of standard_recursion_config: from_values of 135 columns x 2^21, 2^22
    and 2^23 rows at rate 8, cap height 4 (fri/oracle.rs:709-731, 911-977; hash/merkle_tree.rs:283-319). The LDE of the
    last two has 4.5e9 and 9.1e9 elements — more than 2^32, 36 and 72 GB in HBM: every index of the path is 64-bit or
    this fails. 2^21: the C oracle's WHOLE answer (cap and all 2 (2^24 - 16) digests). 2^22, 2^23: three whole columns
    (coefficients and LDE) against the oracle, from_coeffs builds the same tree, 50 opened leaves verify against the cap
    with the oracle's verifier and carry the LDE columns' values, a corrupted leaf is rejected."""
    import time

    import plonky2_gpu_amd as pg
    from plonky2_gpu_amd import _lib

    n_polys, rate_bits, h = 135, 3, 4
    n, n_ext = 1 << log_n, 1 << (log_n + rate_bits)
    assert n_polys * n_ext > (1 << 32) or log_n == 21
    watch = (0, 67, 134)
    whole = log_n == 21
    threads = oracle.usable_threads()
    d_vals = pg.DeviceBuffer(gpu, n_polys * n)
    values = np.empty((len(watch), n), dtype=np.uint64)
    all_values = np.empty((n_polys, n), dtype=np.uint64) if whole else None
    chunk = 5
    for c0 in range(0, n_polys, chunk):
        v = oracle.random_field((chunk, n), seed=9100 + 1000 * log_n + c0)
        d_vals.upload(v, c0 * n)
        if whole:
            all_values[c0:c0 + chunk] = v
        for k, c in enumerate(watch):
            if c0 <= c < c0 + chunk:
                values[k] = v[c - c0]
        del v
    gpu.synchronize()
    t0 = time.perf_counter()
    a = pg.PolynomialBatch.from_values_device(gpu, d_vals, n_polys, log_n, rate_bits, False, h, leaf_major=False)
    gpu.synchronize()
    print("from_values 135 x 2^%d: %.1f ms (first call, tables included)" % (log_n, 1e3 * (time.perf_counter() - t0)))
    cap = a.merkle_tree.cap
    assert cap.shape == (16, 4)
    if whole:
        t0 = time.perf_counter()
        exp = oracle.commit_from_values(all_values, rate_bits, h, threads=threads, want_leaves=False)
        print("oracle commit of 135 x 2^21 on %d threads: %.1f s" % (threads, time.perf_counter() - t0))
        del all_values
        assert (cap == oracle.canon(exp["cap"])).all(), "cap"
        assert (a.merkle_tree.d_digests.download().reshape(-1, 4) == oracle.canon(exp["digests"])).all(), "digests"
        del exp
    # three whole columns: coefficients == the oracle's ifft, LDE column == the oracle's coset LDE in leaf order
    coeffs = oracle.canon(oracle.fft_batch(values, inverse=True, threads=len(watch)))
    del values
    for k, c in enumerate(watch):
        assert (a.d_polynomials.download(c * n, n) == coeffs[k]).all(), ("coefficients", c)
    lde_nat = oracle.canon(oracle.coset_lde_batch(coeffs, rate_bits, threads=len(watch)))
    del coeffs
    perm = bitrev_perm(log_n + rate_bits)
    lde = []
    for k, c in enumerate(watch):
        want = lde_nat[k][perm]
        got = a.d_lde.download(c * n_ext, n_ext)
        assert (got == want).all(), ("LDE", c)
        lde.append(got)
        del want
    del lde_nat, perm
    rng = np.random.default_rng(770 + log_n)
    idx = [0, n_ext - 1, n_ext // 2, n_ext // 2 - 1] + [int(i) for i in rng.integers(0, n_ext, size=46)]
    leaves, sib = a.merkle_tree.open_batch(idx)
    for q, i in enumerate(idx):
        assert oracle.merkle_verify(leaves[q], i, cap, sib[q]), i
        for k, c in enumerate(watch):
            assert leaves[q][c] == lde[k][i], (i, c)
    bad = leaves[7].copy()
    bad[133] ^= np.uint64(1)
    assert not oracle.merkle_verify(bad, idx[7], cap, sib[7])
    del lde
    # from_coeffs on the coefficients: the same tree. The LDE buffer of `a` is released first: two 72 GB LDEs and their
    # trees do not fit beside each other at 2^23 rows.
    digests_a = a.merkle_tree.d_digests
    slots = [0, 1, 2 * (n_ext - 16) - 1] + [int(s) for s in rng.integers(0, 2 * (n_ext - 16), size=200)]
    sampled = [digests_a.download(4 * s, 4) for s in slots]
    d_coeffs = a.d_polynomials
    a.d_lde.free()
    digests_a.free()
    b = pg.PolynomialBatch.from_coeffs_device(gpu, d_coeffs, n_polys, log_n, rate_bits, False, h, leaf_major=False)
    assert (b.merkle_tree.cap == cap).all()
    for s, want in zip(slots, sampled):
        assert (b.merkle_tree.d_digests.download(4 * s, 4) == want).all(), s
    b.d_lde.free()
    b.merkle_tree.d_digests.free()
    d_coeffs.free()


@pytest.mark.parametrize("n_cols,n_rows", [(1, 64), (1, 10), (1, 65), (2, 10), (7, 1000), (64, 4096), (96, 640), (97, 641), (135, 8192), (234, 2048), (300, 129), (20, 1 << 16)])
@pytest.mark.parametrize("kernel", ["strip", "tile"])
def test_leaf_major_copy_and_back(gpu, n_cols, n_rows, kernel):
    """gl_transpose (column-major -> leaf-major; plonky2/src/util/mod.rs:23-53 `transpose`) for whole and ragged strips, one and several
    column chunks, odd and even widths — equal to numpy — with the strip kernels and with the 64 x 64 tiles (PLONKY2_TRANSPOSE=tile, in a
    child process: the choice is read once per process)."""
    import os
    import subprocess
    import sys

    code = f"""
import sys, numpy as np
sys.path.insert(0, {os.path.dirname(os.path.dirname(os.path.abspath(__file__)))!r})
import plonky2_gpu_amd as pg
from plonky2_gpu_amd import _lib
ctx = pg.Context(0)
n_cols, n_rows = {n_cols}, {n_rows}
stride = n_rows + 24
host = np.random.default_rng(n_cols * 7 + n_rows).integers(0, pg.P, size=(n_cols, stride), dtype=np.uint64)
guard = np.full(n_cols * n_rows + 128, 0xDEADBEEFDEADBEEF, dtype=np.uint64)
d_c = pg.DeviceBuffer.from_host(ctx, host); d_r = pg.DeviceBuffer.from_host(ctx, guard)
_lib.call("gl_transpose", d_c.ptr, d_r.ptr, n_cols, n_rows, stride, ctx.ptr); ctx.synchronize()
assert (d_r.download(0, n_rows * n_cols).reshape(n_rows, n_cols) == host[:, :n_rows].T).all()
assert (d_r.download(n_rows * n_cols, 128) == guard[:128]).all(), "wrote past the end of the leaf-major matrix"
print("ok")
"""
    env = dict(os.environ, PLONKY2_TRANSPOSE=kernel)
    if kernel != "strip":  # knobs exist in the diagnostic build only (csrc/knobs.h)
        env["PLONKY2_HIP_LIBRARY"] = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "plonky2_gpu_amd", "libplonky2_hip_debug.so")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout[-1000:] + r.stderr[-2000:]
