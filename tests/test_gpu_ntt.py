"""HIP NTT / inverse NTT / coset LDE vs the CPU oracle (field/src/fft.rs, polynomial/mod.rs).
Integer arithmetic: every comparison is bit-exact on canonical values."""
import numpy as np
import pytest

from gpu_util import P, bitrev_perm, gpu  # noqa: F401

pytestmark = pytest.mark.gpu


def test_fft_rs_fixed_vector(gpu, oracle):
    # field/src/fft.rs:252-282
    import plonky2_gpu_amd as pg

    coeffs = np.array([(i * 1337) % 100 for i in range(200)] + [0] * 56, dtype=np.uint64)
    pts = pg.fft_with_options(gpu, coeffs)
    assert (pts == oracle.canon(oracle.fft(coeffs))).all()
    back = pg.ifft_with_options(gpu, pts)
    assert (back == coeffs).all()


@pytest.mark.parametrize("log_n", list(range(0, 17)))
def test_ntt_matches_oracle_small(gpu, oracle, log_n):
    import plonky2_gpu_amd as pg

    n = 1 << log_n
    n_polys = 5 if log_n < 14 else 3
    x = oracle.random_field((n_polys, n), seed=1000 + log_n)
    x[0, :] = np.uint64(P - 1)  # all -1
    if n > 1:
        x[1, 0] = np.uint64(2**64 - 1)  # non-canonical input representative
    exp_f = oracle.canon(oracle.fft_batch(x))
    exp_i = oracle.canon(oracle.fft_batch(x, inverse=True))
    got_f = pg.fft_with_options(gpu, x)
    assert (got_f == exp_f).all()
    got_i = pg.ifft_with_options(gpu, x)
    assert (got_i == exp_i).all()
    got_b = pg.fft_with_options(gpu, x, bit_reversed=True)
    assert (got_b == exp_f[:, bitrev_perm(log_n)]).all()


@pytest.mark.parametrize("log_n", [17, 18, 19, 20])
def test_ntt_matches_oracle_large(gpu, oracle, log_n):
    import plonky2_gpu_amd as pg

    n = 1 << log_n
    x = oracle.random_field((2, n), seed=2000 + log_n)
    exp_f = oracle.canon(oracle.fft_batch(x, threads=4))
    got_f = pg.fft_with_options(gpu, x)
    assert (got_f == exp_f).all()
    got_i = pg.ifft_with_options(gpu, got_f)
    assert (got_i == x).all()  # ifft(fft(x)) == x, and x is canonical
    got_b = pg.fft_with_options(gpu, x, bit_reversed=True)
    assert (got_b == exp_f[:, bitrev_perm(log_n)]).all()


@pytest.mark.parametrize("log_n,n_polys", [(13, 1500), (14, 700), (16, 150), (20, 40)])
def test_ntt_many_columns_more_workgroups_than_the_chip_holds(gpu, oracle, log_n, n_polys):
    """Batches whose grid exceeds the resident workgroups (2 per CU): the natural-order last pass
    writes transposed, so a transform that kept its intermediate in place would read rows that
    earlier workgroups have already overwritten. Round trip + oracle equality on sampled columns."""
    import plonky2_gpu_amd as pg

    n = 1 << log_n
    x = oracle.random_field((n_polys, n), seed=4000 + log_n)
    f = pg.fft_with_options(gpu, x)
    sample = sorted(set([0, 1, n_polys // 2, n_polys - 2, n_polys - 1]))
    exp = oracle.canon(oracle.fft_batch(x[sample].copy(), threads=4))
    assert (f[sample] == exp).all()
    assert (pg.ifft_with_options(gpu, f) == x).all()
    b = pg.fft_with_options(gpu, x, bit_reversed=True)
    assert (b[sample] == exp[:, bitrev_perm(log_n)]).all()
    assert (b[:, bitrev_perm(log_n)] == f).all()


def test_ntt_strided_batch_in_place(gpu, oracle):
    """polynomials embedded with stride > n (the LDE buffer layout), others untouched."""
    import plonky2_gpu_amd as pg
    from plonky2_gpu_amd import _lib

    log_n, stride, n_polys = 10, 4096, 7
    n = 1 << log_n
    host = oracle.random_field((n_polys, stride), seed=5)
    buf = pg.DeviceBuffer.from_host(gpu, host)
    _lib.call("gl_ntt_batch", buf.ptr, n_polys, log_n, stride, 0, 0, gpu.ptr)
    out = buf.download().reshape(n_polys, stride)
    assert (out[:, :n] == oracle.canon(oracle.fft_batch(host[:, :n].copy()))).all()
    assert (out[:, n:] == host[:, n:]).all()


@pytest.mark.parametrize("log_n,rate_bits,n_polys", [(0, 3, 2), (1, 3, 3), (3, 3, 4), (6, 3, 135), (8, 1, 3), (10, 2, 2),
                                                     (12, 3, 2), (13, 3, 3), (14, 1, 2), (16, 3, 2), (17, 0, 1)])
def test_coset_lde_matches_oracle(gpu, oracle, log_n, rate_bits, n_polys):
    import plonky2_gpu_amd as pg

    n = 1 << log_n
    c = oracle.random_field((n_polys, n), seed=3000 + log_n * 10 + rate_bits)
    got = pg.coset_lde_bit_reversed(gpu, c, rate_bits)
    perm = bitrev_perm(log_n + rate_bits)
    for i in range(n_polys):
        exp = oracle.canon(oracle.coset_lde(c[i], rate_bits))
        assert (got[i] == exp[perm]).all(), i


def test_full_size_round_trip_and_linearity(gpu, oracle):
    """BASELINE config #2 size (2^20) through size-independent properties:
    ifft(fft(x)) == x, fft(a*x + y) == a*fft(x) + fft(y), and oracle equality on 2 columns."""
    import plonky2_gpu_amd as pg

    log_n, n_polys = 20, 8
    n = 1 << log_n
    x = oracle.random_field((n_polys, n), seed=77)
    f = pg.fft_with_options(gpu, x)
    assert (pg.ifft_with_options(gpu, f) == x).all()
    exp = oracle.canon(oracle.fft_batch(x[:2].copy(), threads=2))
    assert (f[:2] == exp).all()
    # linearity with Python ints on a strided sample of outputs
    a = 0x123456789ABCDEF
    z = np.array([[(a * int(u) + int(v)) % P for u, v in zip(x[0, :4096], x[1, :4096])]], dtype=np.uint64)
    zfull = np.zeros((1, n), dtype=np.uint64)
    zfull[0, :4096] = z
    x0 = np.zeros((2, n), dtype=np.uint64)
    x0[0, :4096] = x[0, :4096]
    x0[1, :4096] = x[1, :4096]
    fz = pg.fft_with_options(gpu, zfull)[0]
    f0 = pg.fft_with_options(gpu, x0)
    idx = np.arange(0, n, 4099)
    assert [int(v) for v in fz[idx]] == [(a * int(u) + int(v)) % P for u, v in zip(f0[0, idx], f0[1, idx])]


@pytest.mark.parametrize("log_n", [21, 22, 23, 24])
def test_ntt_three_pass_sizes(gpu, oracle, log_n):
    """2^21 (two passes, 2048-point direct column pass), 2^22 (two passes in every order since round 6: natural forward, the inverse on
    the index-reversed input, bit-reversed in place with the two-waves-per-row kernel for 2048-point rows) and 2^23..2^24 (the
    three-pass plan; 2^24 is the largest size gl_ntt_batch accepts): oracle equality on both columns, round trip, and the bit-reversed
    variant on the second one."""
    import plonky2_gpu_amd as pg

    n = 1 << log_n
    x = oracle.random_field((2, n), seed=5000 + log_n)
    exp = oracle.canon(oracle.fft_batch(x, threads=2))
    f = pg.fft_with_options(gpu, x)
    assert (f == exp).all()
    assert (pg.ifft_with_options(gpu, f) == x).all()
    b = pg.fft_with_options(gpu, x[1], bit_reversed=True)
    assert (b == exp[1][bitrev_perm(log_n)]).all()


def test_two_pass_plan_for_2e22_over_several_workspace_chunks_and_against_the_three_pass_plan(gpu, oracle):
    """2^22 natural-order forward transforms run as 2048 x 2048 in two passes through the workspace (round 5, csrc/ntt.hip): eighteen
    columns = more than the sixteen the 512 MiB workspace holds at once (two chunks), non-canonical inputs in one column; four of them
    (first, the two either side of the chunk boundary, last) against the C oracle, ALL of them against the three-pass plan of the
    diagnostic build (PLONKY2_NTT_TWO_PASS_22=0, a child process). Round 6: the INVERSE takes the same two passes on the index-reversed
    input (csrc/ntt_direct.hip REVIN: the column pass reads x[(n - j) mod n]; the element that wraps, x[0], is one lane's special case,
    per polynomial and per workspace chunk): it brings every column back and equals the three-pass inverse of the diagnostic build on
    input that is NOT a transform of anything (the raw columns x)."""
    import os
    import subprocess
    import sys
    import tempfile

    import plonky2_gpu_amd as pg

    log_n, n_polys = 22, 18
    n = 1 << log_n
    x = oracle.random_field((n_polys, n), seed=2222)
    x[3, :1000] = np.uint64(2**64 - 1) - np.arange(1000, dtype=np.uint64)  # representatives >= p
    f = pg.fft_with_options(gpu, x)
    watch = [0, 15, 16, 17]
    exp = oracle.canon(oracle.fft_batch(x[watch].copy(), threads=4))
    for k, c in enumerate(watch):
        assert (f[c] == exp[k]).all(), c
    assert (pg.ifft_with_options(gpu, f) == oracle.canon(x)).all()
    inv = pg.ifft_with_options(gpu, x)
    brev = pg.fft_with_options(gpu, x, bit_reversed=True)   # two passes in place: column pass + the two-waves-per-row kernel
    assert (brev[17] == f[17][bitrev_perm(log_n)]).all()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.TemporaryDirectory() as tmp:
        np.save(os.path.join(tmp, "x.npy"), x)
        code = f"""
import sys, numpy as np
sys.path.insert(0, {root!r})
import plonky2_gpu_amd as pg
ctx = pg.Context(0)
np.save({os.path.join(tmp, 'f.npy')!r}, pg.fft_with_options(ctx, np.load({os.path.join(tmp, 'x.npy')!r})))
np.save({os.path.join(tmp, 'i.npy')!r}, pg.ifft_with_options(ctx, np.load({os.path.join(tmp, 'x.npy')!r})))
np.save({os.path.join(tmp, 'b.npy')!r}, pg.fft_with_options(ctx, np.load({os.path.join(tmp, 'x.npy')!r}), bit_reversed=True))
"""
        env = dict(os.environ, PLONKY2_NTT_TWO_PASS_22="0", PLONKY2_NTT_TWO_PASS_22_INPLACE="0", PLONKY2_HIP_LIBRARY=os.path.join(root, "plonky2_gpu_amd", "libplonky2_hip_debug.so"))
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        assert (np.load(os.path.join(tmp, "f.npy")) == f).all(), "the two-pass and the three-pass plan disagree"
        assert (np.load(os.path.join(tmp, "i.npy")) == inv).all(), "the two-pass and the three-pass INVERSE disagree"
        assert (np.load(os.path.join(tmp, "b.npy")) == brev).all(), "the two-pass and the three-pass BIT-REVERSED transform disagree"


def test_ntt_batch_with_more_than_2e32_elements(gpu, oracle):
    """gl_ntt_batch over 520 polynomials of 2^23 points = 4.36e9 elements (35 GB) in ONE call: every polynomial offset is 64-bit or
    the later ones land on the earlier. The buffer is filled on the device with copies of eight distinct columns; natural order (the
    workspace holds eight polynomials at a time: 65 chunks) and bit-reversed (in place); polynomials 0, 7, 263 and 519 against the C
    oracle, all replicas of one column against each other through a device-side comparison."""
    import plonky2_gpu_amd as pg
    from plonky2_gpu_amd import _lib

    log_n, distinct, copies = 23, 8, 65
    n, n_polys = 1 << log_n, 8 * 65
    assert n_polys * n > 1 << 32
    x = oracle.random_field((distinct, n), seed=2323)
    exp = oracle.canon(oracle.fft_batch(x, threads=8))
    perm = bitrev_perm(log_n)
    buf = pg.DeviceBuffer(gpu, n_polys * n)
    for order, expect in ((0, exp), (1, exp[:, perm])):
        buf.upload(x, 0)
        for k in range(1, copies):
            _lib.call("gl_memcpy_d2d", buf.at(k * distinct * n), buf.ptr, 8 * distinct * n, gpu.ptr)
        _lib.call("gl_ntt_batch", buf.ptr, n_polys, log_n, n, 0, order, gpu.ptr)
        gpu.synchronize()
        for c in (0, 7, 263, 519):
            assert (buf.download(c * n, n) == expect[c % distinct]).all(), (order, c)
        # every replica equals the first copy: subtract block k from block 0 on the device (op 1 = sub), the result must be all zero
        diff = pg.DeviceBuffer(gpu, distinct * n)
        for k in (1, 31, 32, 33, 64):
            _lib.call("gl_debug_field_op", 1, buf.ptr, buf.at(k * distinct * n), diff.ptr, distinct * n, gpu.ptr)
            assert not diff.download().any(), (order, k)
        diff.free()
    buf.free()


@pytest.mark.parametrize("log_n,rate_bits", [(21, 1), (22, 1), (23, 0), (21, 3), (24, 0)])
def test_coset_lde_three_pass(gpu, oracle, log_n, rate_bits):
    """Coset LDE of the large sizes: 2^21 in two passes (split 2048-point columns), 2^22 .. 2^24 points in three."""
    import plonky2_gpu_amd as pg

    c = oracle.random_field((2, 1 << log_n), seed=6000)
    got = pg.coset_lde_bit_reversed(gpu, c, rate_bits)
    perm = bitrev_perm(log_n + rate_bits)
    for i in range(2):
        assert (got[i] == oracle.canon(oracle.coset_lde(c[i], rate_bits))[perm]).all()


@pytest.mark.parametrize("log_n", [0, 3, 8, 12, 15, 21])
def test_coset_fft_and_ifft_natural_order(gpu, oracle, log_n):
    # field/src/polynomial/mod.rs:482-522 (random poly, shift = generator) and a second shift
    import plonky2_gpu_amd as pg

    x = oracle.random_field((2, 1 << log_n), seed=7000 + log_n)
    for shift in (7, 0x1234567890ABCDEF):
        ev = pg.coset_fft(gpu, x, shift)
        for i in range(2):
            assert (ev[i] == oracle.canon(oracle.coset_fft(x[i], shift))).all()
        assert (pg.coset_ifft(gpu, ev, shift) == x).all()
        for i in range(2):
            assert (pg.coset_ifft(gpu, x[i], shift) == oracle.canon(oracle.coset_ifft(x[i], shift))).all()


def test_more_distinct_cosets_than_the_table_cache_holds(gpu, oracle):
    """The per-device coset-table cache (capi.hip get_coset_tables) is a bounded LRU: a process that has used
    more distinct (size, rate, shift) combinations than it holds must keep working and keep being right,
    including for a combination that was evicted and is built again. 150 shifts x (forward + inverse tables)
    is several times the bound whatever ran earlier in this process."""
    import plonky2_gpu_amd as pg

    log_n = 6
    x = oracle.random_field((1 << log_n,), seed=7100)
    shifts = [7] + [int(s) for s in oracle.random_field((149,), seed=7101) if int(s) != 0]
    first = pg.coset_fft(gpu, x, shifts[0])
    assert (first == oracle.canon(oracle.coset_fft(x, shifts[0]))).all()
    for shift in shifts[1:]:
        ev = pg.coset_fft(gpu, x, shift)
        assert (ev == oracle.canon(oracle.coset_fft(x, shift))).all(), shift
        assert (pg.coset_ifft(gpu, ev, shift) == x).all(), shift
    assert (pg.coset_fft(gpu, x, shifts[0]) == first).all()  # long since evicted: rebuilt, same answer
    # and a combination of the LDE kind after the churn
    c = oracle.random_field((1 << 10,), seed=7102)
    got = pg.coset_lde_bit_reversed(gpu, c.reshape(1, -1), 2)
    assert (got[0] == oracle.canon(oracle.coset_lde(c, 2))[bitrev_perm(12)]).all()


@pytest.mark.parametrize(
    "env",
    [
        {"PLONKY2_NTT_KERNEL": "tile"},  # workgroup-tile kernel everywhere, three passes from 2^21
        {"PLONKY2_NTT_WIDE": "0"},  # 8192-element tiles only: no 128-byte column pass, no split columns (2^21 in three passes)
        {"PLONKY2_NTT_XCD": "0", "PLONKY2_NTT_WG_PER_CU": "1"},  # plain tile order, one workgroup per CU
        {"PLONKY2_NTT_DIRECT": "0"},  # the wave-tile kernels also where a direct pass exists (round 2's kernels for 2^16 - 2^20)
        {"PLONKY2_NTT_DIRECT": "0", "PLONKY2_NTT_CHUNK_COLS": "1"},  # natural order through the workspace one column at a time
    ],
    ids=["tile-kernel", "narrow-tiles", "plain-order", "wave-tiles", "one-column-chunks"],
)
def test_alternative_kernel_selections(gpu, env):
    """The DIAGNOSTIC build of the library (csrc/knobs.h, libplonky2_hip_debug.so) picks its pass kernels once per process
    from the environment; every selection must give the oracle's results: the kernels that the product library uses only
    for ragged tiles, long columns or sizes without a direct pass stay under test at every size. Runs
    tests/ntt_variant_child.py under each."""
    import os
    import subprocess
    import sys

    child = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ntt_variant_child.py")
    debug_lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "plonky2_gpu_amd", "libplonky2_hip_debug.so")
    assert os.path.exists(debug_lib), "make -C plonky2_gpu_amd/csrc debug (done by __graft_entry__.build())"
    r = subprocess.run([sys.executable, child], env=dict(os.environ, PLONKY2_HIP_LIBRARY=debug_lib, **env), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout[-2000:] + r.stderr[-2000:]


def test_ntt_argument_errors(gpu):
    import plonky2_gpu_amd as pg
    from plonky2_gpu_amd import _lib

    buf = pg.DeviceBuffer(gpu, 1 << 10)
    with pytest.raises(pg.Plonky2HipError) as e:
        _lib.call("gl_ntt_batch", buf.ptr, 1, 25, 1 << 25, 0, 0, gpu.ptr)
    assert e.value.code == pg.GL_E_INVALID
    with pytest.raises(pg.Plonky2HipError):
        _lib.call("gl_ntt_batch", buf.ptr, 1, 10, 512, 0, 0, gpu.ptr)  # stride < n
    with pytest.raises(pg.Plonky2HipError):
        _lib.call("gl_ntt_batch", buf.ptr, 1, 4, 16, 1, 1, gpu.ptr)  # bit-reversed inverse
    with pytest.raises(ValueError):
        pg.fft_with_options(gpu, np.zeros(12, dtype=np.uint64))  # not a power of two
    _lib.call("gl_ntt_batch", buf.ptr, 0, 10, 1024, 0, 0, gpu.ptr)  # empty batch is a no-op


def test_sizes_above_the_limit_are_rejected(gpu):
    """log_n <= 24 is the documented limit of gl_ntt_batch / gl_coset_lde_batch (include/plonky2_hip.h): beyond it an error, not a wrong answer"""
    import plonky2_gpu_amd as pg
    from plonky2_gpu_amd import _lib

    buf = pg.DeviceBuffer(gpu, 16)
    with pytest.raises(pg.Plonky2HipError) as e:
        _lib.call("gl_ntt_batch", buf.ptr, 1, 25, 1 << 25, 0, 0, gpu.ptr)
    assert e.value.code == pg.GL_E_INVALID
    with pytest.raises(pg.Plonky2HipError) as e:
        _lib.call("gl_coset_lde_batch", buf.ptr, buf.ptr, 1, 25, 0, 7, 1 << 25, 1 << 25, gpu.ptr)
    assert e.value.code == pg.GL_E_INVALID
    buf.free()


@pytest.mark.parametrize("log_n", [12, 16, 20, 21])
def test_representatives_that_fire_the_deferred_rare_paths(gpu, oracle, log_n):
    """The passes' field operations defer their rare corrections (csrc/gl_field.h: add's second wrap needs both operands >= p, sub's
    second borrow a subtrahend > p and a minuend below 2^32) — events of probability 2^-64 on canonical random data. Here they fire
    in most lanes of the first radix stage: the same field elements given as NON-CANONICAL representatives (x + p wherever that fits
    64 bits, i.e. for x < 2^32 - 1) mixed with tiny values. Every transform of them must equal the transform of the canonical
    vector: natural, inverse, bit-reversed, and the coset LDE. (A build whose rare_any() always says "no" fails all four sizes of this
    test and tests/test_gpu_field.py::test_deferred_rare_paths: checked once by hand, round 4.)"""
    import plonky2_gpu_amd as pg

    n = 1 << log_n
    rng = np.random.default_rng(7000 + log_n)
    small = rng.integers(0, (1 << 32) - 1, size=(2, n), dtype=np.uint64)        # all representable as x + p
    tiny = rng.integers(0, 4, size=(2, n), dtype=np.uint64)
    pick = rng.integers(0, 4, size=(2, n))
    canonical = np.where(pick == 0, tiny, small).astype(np.uint64)
    lifted = np.where(pick >= 2, canonical + np.uint64(P), canonical)            # half of the entries >= p, the rest < 2^32
    assert (lifted >= np.uint64(P)).sum() > n // 2 and ((lifted % np.uint64(P)) == canonical).all()
    exp = oracle.canon(oracle.fft_batch(canonical, threads=2))
    assert (pg.fft_with_options(gpu, lifted) == exp).all()
    assert (pg.fft_with_options(gpu, lifted[1], bit_reversed=True) == exp[1][bitrev_perm(log_n)]).all()
    assert (pg.ifft_with_options(gpu, lifted) == pg.ifft_with_options(gpu, canonical)).all()
    if log_n <= 20:
        a = pg.coset_lde_bit_reversed(gpu, lifted, 3)
        b = pg.coset_lde_bit_reversed(gpu, canonical, 3)
        assert (a == b).all()
