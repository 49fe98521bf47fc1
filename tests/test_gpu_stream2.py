"""The one overlap the reference's caller relies on (SURVEY §8b "Threading / sync"), through the C ABI.

PolynomialBatch::from_values_with_gpu (plonky2/src/fri/oracle.rs:394-422): `ifft` on region A, then an ASYNC device-to-host copy
of the coefficients queued on ctx->stream2 (:403-407, rustacuda async_copy_to), then `merkle_tree_from_coeffs(values_device,
values_device, ..)` -- region A passed both as the coefficients and as the place of the leaf-major LDE -- and right after the
call returns the host READS the copy's destination (:462). from_coeffs_with_gpu does the same without the ifft (:595-640).
The reference's body honours this with cudaStreamSynchronize(ctx->stream2) before its transposition overwrites region A
(cuda/plonky2_gpu.cu:586). Here the order is an event recorded on stream2 that the hashing / transposing stream waits for
(csrc/capi.hip commit_from_coeffs_impl) plus a final wait for stream2 in the symbol.

stream2 is kept busy with earlier copies so that, without that order, the leaves WOULD be written long before the copy of the
coefficients runs: with PLONKY2_DROP_STREAM2_WAIT=1 in the diagnostic build these tests fail (tools/gpu_runs/stream2_negative.sh
records it); in the product they must pass."""
import ctypes
import os

import numpy as np
import pytest

from gpu_util import gpu  # noqa: F401

P = 0xFFFFFFFF00000001
hipMemcpyDeviceToHost = 2


def _hip():
    for name in ("libamdhip64.so", "/opt/rocm/lib/libamdhip64.so"):
        try:
            lib = ctypes.CDLL(name)
            break
        except OSError:
            continue
    else:
        pytest.fail("libamdhip64.so not found")
    lib.hipMemcpyAsync.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
    lib.hipMemcpyAsync.restype = ctypes.c_int
    return lib


@pytest.mark.gpu
@pytest.mark.parametrize("from_values", [True, False])
@pytest.mark.parametrize("polys,log_n", [(80, 14), (20, 9)])  # the pipelined commit (>= 48 columns, n_ext >= 2^16) and the one-shot one
def test_copy_on_stream2_survives_the_aliased_commit(gpu, oracle, from_values, polys, log_n):
    import plonky2_gpu_amd as pg
    from plonky2_gpu_amd import _lib

    hip = _hip()
    rate_bits, h = 3, 4
    n, n_ext = 1 << log_n, 1 << (log_n + rate_bits)
    pad, nd = polys * n_ext, 2 * (n_ext - (1 << h))
    total = 2 * pad + 4 * nd + (4 << h)
    vals = oracle.random_field((polys, n), seed=1200 + polys + from_values)
    exp = oracle.commit_from_values(vals, rate_bits, h, threads=4) if from_values else oracle.commit_from_coeffs(vals, rate_bits, h, threads=4)
    coeffs = oracle.canon(exp["coeffs"]) if from_values else vals
    stream2 = ctypes.cast(gpu.ptr, ctypes.POINTER(ctypes.c_void_p))[1]  # ctx = {stream, stream2}, the twin of CudaInnerContext
    region = pg.DeviceBuffer(gpu, total)
    region.upload(vals, 0)
    host_copy = pg.PinnedArray(polys * n)
    host_copy.array[:] = 0
    busy_src = pg.DeviceBuffer(gpu, 1 << 24)  # 128 MiB, unrelated to the commit
    busy_dst = pg.PinnedArray(1 << 24)
    if from_values:
        n_inv = ctypes.c_uint64(P - ((P - 1) >> log_n))
        _lib.call("ifft", region.ptr, polys, n, log_n, None, ctypes.addressof(n_inv), gpu.ptr)
    gpu.synchronize()
    for _ in range(6):  # ~15 ms of copies ahead of the one that matters: the commit itself takes about 1 ms
        assert hip.hipMemcpyAsync(busy_dst.ptr, busy_src.ptr, busy_src.n * 8, hipMemcpyDeviceToHost, stream2) == 0
    assert hip.hipMemcpyAsync(host_copy.ptr, region.ptr, polys * n * 8, hipMemcpyDeviceToHost, stream2) == 0
    _lib.call("merkle_tree_from_coeffs", region.ptr, region.ptr, polys, n, log_n, None, None, None, rate_bits, 0, h, pad, gpu.ptr)
    # no synchronisation by the caller here: the reference's caller reads values_flatten as soon as the call has returned
    got_copy = host_copy.array.copy()
    assert (got_copy.reshape(polys, n) == coeffs).all(), "the coefficients copied on stream2 were overwritten or not complete"
    assert (region.download(2 * pad + 4 * nd, 4 << h).reshape(-1, 4) == oracle.canon(exp["cap"])).all()
    assert (region.download(2 * pad, 4 * nd).reshape(-1, 4) == oracle.canon(exp["digests"])).all()
    assert (region.download(0, pad).reshape(n_ext, polys) == oracle.canon(exp["leaves"])).all(), "leaf-major LDE in region A"
    assert (region.download(pad, pad).reshape(polys, n_ext) == oracle.canon(exp["leaves"]).T).all(), "column-major LDE in region B"
    for b in (region, busy_src):
        b.free()
    for b in (host_copy, busy_dst):
        b.free()
