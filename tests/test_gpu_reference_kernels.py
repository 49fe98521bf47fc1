"""The product against the REFERENCE'S OWN device kernels, run on the same MI355X (oracle/_ref, see oracle/ref_harness.hip):
cuda/plonky2_gpu_impl.cuh compiled unmodified for gfx950, launched with the geometry of cuda/plonky2_gpu.cu. Same device
buffers through the reference kernel and through the product's C-ABI entry point, bit-exact after canonicalisation.

  reference kernel(s)                                               product entry point                    size
  ifft_kernel :214 / fft_kernel :254                                ifft (symbol), gl_ntt_batch            2^18
  lde_kernel :260 + init_lde + mul_shift :299 + fft_kernel(r)       gl_coset_lde_batch                     2^15 -> 2^18
  hash_leaves_kernel :349 + reduce_digests_kernel :411              gl_merkle_tree_from_columns            2^16 leaves x 135
  all of merkle_tree_from_coeffs (plonky2_gpu.cu:435-606)           merkle_tree_from_coeffs (symbol)       135 x 2^13, whole region
  compute_quotient_values_kernel :485 (+ transpose, ifft, mul)      compute_quotient_polys (symbol)        log_len 4, 8, 18

These are the reference's GPU twin of its CPU prover (the code its authors produced proofs with), not the Rust itself; they
are what pins rows a12 (quotient: gate constraints of all 25 gates, filters, alpha-reduction order, partial-product checks,
L_0, 1/Z_H) and, at log_len 4, the Python oracle of that row (oracle/plonk_ref.py + gates_ref.py) as well.

When oracle/_ref/libplonky2_ref.so is absent these tests SKIP LOUDLY (the reason names the build step)."""
import ctypes
import os
import sys

import numpy as np
import pytest

from gpu_util import bitrev_perm, gpu  # noqa: F401

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
P = 0xFFFFFFFF00000001


@pytest.fixture(scope="module")
def ref():
    from oracle import ref_gpu

    if not ref_gpu.available():
        pytest.skip("REFERENCE KERNELS NOT COMPARED: " + ref_gpu.why_absent())
    ref_gpu.lib()
    return ref_gpu


@pytest.fixture(scope="module")
def o():
    from oracle import oracle

    oracle.build()
    return oracle


def canon(a):
    a = np.asarray(a, dtype=np.uint64)
    return np.where(a >= np.uint64(P), a - np.uint64(P), a)


def powers(base, count):
    """[base^0 .. base^(count-1)] mod p by doubling, vectorised (tools/synth_circuit.py's numpy field arithmetic)"""
    import synth_circuit as sc

    out = np.ones(1, dtype=np.uint64)
    b = base % P
    while out.size < count:
        out = np.concatenate([out, sc.np_mul(out, np.uint64(pow(b, out.size, P)))])
    return np.ascontiguousarray(out[:count])


def rnd(rng, *shape):
    return rng.integers(0, P, size=shape, dtype=np.uint64)


@pytest.mark.gpu
def test_ifft_and_fft_kernels_at_2_18(gpu, ref, o):
    import plonky2_gpu_amd as pg
    from plonky2_gpu_amd import _lib

    log_n, polys = 18, 5
    n = 1 << log_n
    x = rnd(np.random.default_rng(418), polys, n)
    table = pg.DeviceBuffer.from_host(gpu, o.root_table_concat(n))
    n_inv = ref.n_inv(log_n)
    # inverse: the reference's `ifft` launch against the product's `ifft` symbol and gl_ntt_batch
    a = pg.DeviceBuffer.from_host(gpu, x)
    ref.call("ref_ifft", a.ptr, polys, n, log_n, table.ptr, n_inv)
    exp = canon(a.download()).reshape(polys, n)
    b = pg.DeviceBuffer.from_host(gpu, x)
    h_inv = ctypes.c_uint64(n_inv)
    _lib.call("ifft", b.ptr, polys, n, log_n, table.ptr, ctypes.addressof(h_inv), gpu.ptr)
    assert (b.download().reshape(polys, n) == exp).all()
    assert (pg.ifft_with_options(gpu, x) == exp).all()
    assert (exp[0] == o.canon(o.ifft(x[0]))).all()  # and the C restatement of fft.rs agrees with the reference kernel
    # forward, natural order (fft_kernel with r = 0) and with a zero factor r = 3 on a zero-padded input
    a.upload(x)
    ref.call("ref_fft", a.ptr, polys, n, log_n, table.ptr, 0)
    fwd = canon(a.download()).reshape(polys, n)
    assert (pg.fft_with_options(gpu, x) == fwd).all()
    assert (fwd[1] == o.canon(o.fft(x[1]))).all()
    padded = x.copy()
    padded[:, n >> 3:] = 0
    # the kernel's r: input element i lives at i << r after the bit reversal, i.e. the first n/8 coefficients, rest zero
    a.upload(padded)
    ref.call("ref_fft", a.ptr, polys, n, log_n, table.ptr, 3)
    assert (pg.fft_with_options(gpu, padded) == canon(a.download()).reshape(polys, n)).all()
    for buf in (a, b, table):
        buf.free()


@pytest.mark.gpu
def test_coset_lde_kernels_2_15_to_2_18(gpu, ref, o):
    """cuda/plonky2_gpu.cu:481-547 (copy, zero, scale by 7^i, fft with r = rate_bits, bit reversal) = gl_coset_lde_batch"""
    import plonky2_gpu_amd as pg
    from plonky2_gpu_amd import _lib

    log_n, rate_bits, polys = 15, 3, 7
    n, n_ext = 1 << log_n, 1 << (log_n + rate_bits)
    coeffs = rnd(np.random.default_rng(515), polys, n)
    d_coeffs = pg.DeviceBuffer.from_host(gpu, coeffs)
    table2 = pg.DeviceBuffer.from_host(gpu, o.root_table_concat(n_ext))
    shifts = pg.DeviceBuffer.from_host(gpu, powers(7, n))
    d_ref = pg.DeviceBuffer(gpu, polys * n_ext)
    ref.call("ref_coset_lde", d_coeffs.ptr, d_ref.ptr, polys, n, log_n, table2.ptr, shifts.ptr, rate_bits, n_ms=4)
    natural = canon(d_ref.download()).reshape(polys, n_ext)
    assert (natural[2] == o.canon(o.coset_lde(coeffs[2], rate_bits))).all()
    ref.call("ref_reverse_index_bits", d_ref.ptr, polys, n_ext, log_n + rate_bits)
    exp = canon(d_ref.download()).reshape(polys, n_ext)
    assert (exp == natural[:, bitrev_perm(log_n + rate_bits)]).all()
    d_out = pg.DeviceBuffer(gpu, polys * n_ext)
    _lib.call("gl_coset_lde_batch", d_coeffs.ptr, d_out.ptr, polys, log_n, rate_bits, 7, n, n_ext, gpu.ptr)
    assert (d_out.download().reshape(polys, n_ext) == exp).all()
    for buf in (d_coeffs, table2, shifts, d_ref, d_out):
        buf.free()


@pytest.mark.gpu
@pytest.mark.parametrize("leaf_len,log_leaves,cap_height", [(135, 16, 4), (20, 12, 0), (9, 10, 1)])
def test_merkle_kernels(gpu, ref, leaf_len, log_leaves, cap_height):
    """hash_leaves_kernel + reduce_digests_kernel: digests in the reference's recursive layout (its find_digest_index) and the cap"""
    import plonky2_gpu_amd as pg
    from plonky2_gpu_amd import _lib

    n = 1 << log_leaves
    nd = 2 * (n - (1 << cap_height))
    cols = rnd(np.random.default_rng(616 + leaf_len), leaf_len, n)
    region = pg.DeviceBuffer(gpu, leaf_len * n + 4 * nd + (4 << cap_height))
    region.upload(cols, 0)
    ref.call("ref_merkle_tree", region.ptr, leaf_len, n, cap_height, n_ms=2)
    exp_dig = canon(region.download(leaf_len * n, 4 * nd))
    exp_cap = canon(region.download(leaf_len * n + 4 * nd, 4 << cap_height))
    d_dig, d_cap = pg.DeviceBuffer(gpu, max(4 * nd, 1)), pg.DeviceBuffer(gpu, 4 << cap_height)
    _lib.call("gl_merkle_tree_from_columns", region.ptr, leaf_len, n, n, cap_height, d_dig.ptr, d_cap.ptr, gpu.ptr)
    assert (d_cap.download() == exp_cap).all()
    assert (d_dig.download()[:4 * nd] == exp_dig).all()
    for buf in (region, d_dig, d_cap):
        buf.free()


@pytest.mark.gpu
def test_whole_merkle_tree_from_coeffs_region(gpu, ref, o):
    """Every kernel launch of the reference's merkle_tree_from_coeffs (plonky2_gpu.cu:435-606) in order, against the product's symbol of
    that name: leaf-major LDE at ext[0..], column-major bit-reversed LDE at ext[pad..], digests || cap behind it."""
    import plonky2_gpu_amd as pg
    from plonky2_gpu_amd import _lib

    polys, log_n, rate_bits, h = 135, 13, 3, 4
    n, n_ext = 1 << log_n, 1 << (log_n + rate_bits)
    pad, nd = polys * n_ext, 2 * (n_ext - (1 << h))
    total = 2 * pad + 4 * nd + (4 << h)
    coeffs = rnd(np.random.default_rng(717), polys, n)
    table2 = pg.DeviceBuffer.from_host(gpu, o.root_table_concat(n_ext))
    shifts = pg.DeviceBuffer.from_host(gpu, powers(7, n))
    a = pg.DeviceBuffer(gpu, total)
    a.upload(coeffs, 0)
    ref.call("ref_coset_lde", a.ptr, a.at(pad), polys, n, log_n, table2.ptr, shifts.ptr, rate_bits, n_ms=4)
    ref.call("ref_reverse_index_bits", a.at(pad), polys, n_ext, log_n + rate_bits)
    ref.call("ref_merkle_tree", a.at(pad), polys, n_ext, h, n_ms=2)
    ref.call("ref_transpose", a.at(pad), a.ptr, polys, n_ext)
    exp = canon(a.download())
    b = pg.DeviceBuffer(gpu, total)
    b.upload(coeffs, 0)
    _lib.call("merkle_tree_from_coeffs", b.ptr, b.ptr, polys, n, log_n, None, table2.ptr, shifts.ptr, rate_bits, 0, h, pad, gpu.ptr)
    got = b.download()
    assert (got[2 * pad + 4 * nd:] == exp[2 * pad + 4 * nd:]).all(), "cap"
    assert (got[2 * pad:2 * pad + 4 * nd] == exp[2 * pad:2 * pad + 4 * nd]).all(), "digests"
    assert (got[pad:2 * pad] == exp[pad:2 * pad]).all(), "column-major LDE"
    assert (got[:pad] == exp[:pad]).all(), "leaf-major LDE"
    for buf in (a, b, table2, shifts):
        buf.free()


# ---- the quotient --------------------------------------------------------------------------------------------------------


def quotient_instance(log_len, seed):
    from test_reference_quotient import random_instance

    return random_instance(log_len, seed)


def reference_quotient(gpu, ref, o, inst, bufs, pih, whole):
    """whole: the reference's full launch sequence -> coefficient polynomials [2][n_ext]; else only the per-point kernel, then the
    coset iFFT by the C oracle (ifft_kernel asserts n_ext > 256)"""
    import plonky2_gpu_amd as pg
    from oracle import pyref
    from plonky2_gpu_amd import ed25519_circuit as ed

    log_len, n_ext = inst["log_len"], inst["n_ext"]
    n, bits = 1 << log_len, log_len + ed.RATE_BITS
    w = pyref.root_of_unity(bits)
    points = pg.DeviceBuffer.from_host(gpu, powers(w, n_ext))
    g_pow_n = pow(ed.COSET_SHIFT, n, P)
    w_rate = pyref.root_of_unity(ed.RATE_BITS)
    zh = [(g_pow_n * pow(w_rate, i, P) - 1) % P for i in range(1 << ed.RATE_BITS)]  # ZeroPolyOnCoset::new, zero_poly_coset.rs:20-37
    zh_ev = pg.DeviceBuffer.from_host(gpu, np.array(zh, dtype=np.uint64))
    zh_inv = pg.DeviceBuffer.from_host(gpu, np.array([pow(v, P - 2, P) for v in zh], dtype=np.uint64))
    outs = pg.DeviceBuffer(gpu, 2 * n_ext)
    h_pih = (ctypes.c_uint64 * 4)(*[int(v) for v in pih])
    common = (bufs["wires"].ptr, log_len, ed.RATE_BITS, bufs["zs"].ptr, bufs["cs"].ptr, outs.ptr)
    tail = (points.ptr, zh_ev.ptr, zh_inv.ptr, bufs["k_is"].ptr, bufs["alphas"].ptr, bufs["betas"].ptr, bufs["gammas"].ptr)
    if whole:
        polys = pg.DeviceBuffer(gpu, 2 * n_ext)
        table2 = pg.DeviceBuffer.from_host(gpu, o.root_table_concat(n_ext))
        shift_inv = pg.DeviceBuffer.from_host(gpu, powers(pow(ed.COSET_SHIFT, P - 2, P), n_ext))
        ms = ref.call("ref_compute_quotient_polys", *common, polys.ptr, *tail, table2.ptr, shift_inv.ptr, h_pih, ref.n_inv(bits), n_ms=4)
        out = canon(polys.download()).reshape(2, n_ext)
        for buf in (polys, table2, shift_inv):
            buf.free()
    else:
        ms = ref.call("ref_compute_quotient_values", *common, *tail, h_pih)
        vals = canon(outs.download()).reshape(n_ext, 2)
        out = np.stack([o.canon(o.coset_ifft(np.ascontiguousarray(vals[:, c]), ed.COSET_SHIFT)) for c in range(2)])
    for buf in (points, zh_ev, zh_inv, outs):
        buf.free()
    return out, ms


@pytest.mark.gpu
def test_quotient_kernel_pins_the_python_oracle_and_the_symbol_at_log_len_4(gpu, ref, o):
    """three ways on 2^7 random points where every gate is live: the reference's kernel, the product's symbol, oracle/plonk_ref.py"""
    from plonky2_gpu_amd import ed25519_circuit as ed
    from test_reference_quotient import oracle_quotient, run_symbol

    inst = quotient_instance(4, seed=9404)
    got, bufs = run_symbol(gpu, inst)
    exp, _ = reference_quotient(gpu, ref, o, inst, bufs, ed.REFERENCE_PUBLIC_INPUTS_HASH, whole=False)
    assert (got == exp).all(), "product symbol vs the reference's kernel"
    py = np.array(oracle_quotient(inst, ed.REFERENCE_PUBLIC_INPUTS_HASH), dtype=np.uint64)
    assert (py == exp).all(), "oracle/plonk_ref.py + gates_ref.py vs the reference's kernel"


@pytest.mark.gpu
def test_quotient_launch_sequence_at_log_len_8(gpu, ref, o):
    """the full sequence (kernel, transpose_kernel, ifft_kernel, mul_kernel), with another proof's public-inputs hash"""
    import plonky2_gpu_amd as pg
    from test_reference_quotient import run_symbol

    inst = quotient_instance(8, seed=9408)
    other = [int(v) for v in rnd(np.random.default_rng(3), 4)]
    try:
        pg.reference_set_public_inputs_hash(other)
        got, bufs = run_symbol(gpu, inst)
    finally:
        pg.reference_set_public_inputs_hash(None)
    exp, _ = reference_quotient(gpu, ref, o, inst, bufs, other, whole=True)
    assert (got == exp).all()
    vals_route, _ = reference_quotient(gpu, ref, o, inst, bufs, other, whole=False)
    assert (vals_route == exp).all()


@pytest.mark.gpu
def test_quotient_at_the_size_the_reference_hard_wires(gpu, ref, o):
    """log_len 18 (2^21 LDE points, 5.7 GB of random leaves, all 25 gates live at every point): every coefficient of both
    quotient polynomials equal to what the reference's own launch sequence produces on the same device buffers."""
    from plonky2_gpu_amd import ed25519_circuit as ed
    from test_reference_quotient import run_symbol

    inst = quotient_instance(18, seed=9200)
    got, bufs = run_symbol(gpu, inst)
    exp, ms = reference_quotient(gpu, ref, o, inst, bufs, ed.REFERENCE_PUBLIC_INPUTS_HASH, whole=True)
    print("reference kernels on this device, log_len 18: quotient values %.1f ms, transpose %.2f, ifft %.1f, mul %.2f" % tuple(ms))
    assert got.shape == exp.shape == (2, 1 << 21)
    assert (got == exp).all()
