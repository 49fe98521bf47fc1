"""Host-side mirror of field/src/fft.rs for batches resident in HBM."""
import numpy as np

from . import _lib
from .device import DeviceBuffer


def _as_batch(values):
    a = np.ascontiguousarray(values, dtype=np.uint64)
    if a.ndim == 1:
        a = a[None, :]
    n = a.shape[1]
    if n & (n - 1) or n == 0:
        raise ValueError("polynomial length must be a power of two")  # log2_strict panics in the reference
    return a, n.bit_length() - 1


def fft_with_options(ctx, values, zero_factor=None, root_table=None, bit_reversed=False):
    """fft_with_options (field/src/fft.rs:58-66) on each row of `values` ([n_polys, n] or [n]).

    `zero_factor` and `root_table` are accepted for signature parity; they are performance hints
    in the reference (fft.rs:203-217) and do not change the result."""
    a, log_n = _as_batch(values)
    buf = DeviceBuffer.from_host(ctx, a)
    _lib.call("gl_ntt_batch", buf.ptr, a.shape[0], log_n, a.shape[1], 0, int(bit_reversed), ctx.ptr)
    out = buf.download().reshape(a.shape)
    buf.free()
    return out if np.ndim(values) > 1 else out[0]


def ifft_with_options(ctx, values, zero_factor=None, root_table=None):
    """ifft_with_options (field/src/fft.rs:73-103)."""
    a, log_n = _as_batch(values)
    buf = DeviceBuffer.from_host(ctx, a)
    _lib.call("gl_ntt_batch", buf.ptr, a.shape[0], log_n, a.shape[1], 1, 0, ctx.ptr)
    out = buf.download().reshape(a.shape)
    buf.free()
    return out if np.ndim(values) > 1 else out[0]


def coset_lde_bit_reversed(ctx, coeffs, rate_bits, shift=7):
    """lde(rate_bits).coset_fft_with_options(shift, Some(rate_bits)) per row
    (field/src/polynomial/mod.rs:205-207, 286-299), returned in bit-reversed (leaf) order."""
    a, log_n = _as_batch(coeffs)
    n_ext = a.shape[1] << rate_bits
    src = DeviceBuffer.from_host(ctx, a)
    dst = DeviceBuffer(ctx, a.shape[0] * n_ext)
    _lib.call("gl_coset_lde_batch", src.ptr, dst.ptr, a.shape[0], log_n, rate_bits, shift, a.shape[1], n_ext, ctx.ptr)
    out = dst.download().reshape(a.shape[0], n_ext)
    src.free()
    dst.free()
    return out if np.ndim(coeffs) > 1 else out[0]


def coset_fft(ctx, coeffs, shift=7):
    """PolynomialCoeffs::coset_fft(shift) (field/src/polynomial/mod.rs:281-299), natural order."""
    a, log_n = _as_batch(coeffs)
    buf = DeviceBuffer.from_host(ctx, a)
    _lib.call("gl_coset_ntt_batch", buf.ptr, a.shape[0], log_n, a.shape[1], shift, 0, ctx.ptr)
    out = buf.download().reshape(a.shape)
    buf.free()
    return out if np.ndim(coeffs) > 1 else out[0]


def coset_ifft(ctx, values, shift=7):
    """PolynomialValues::coset_ifft(shift) (field/src/polynomial/mod.rs:64-77)."""
    a, log_n = _as_batch(values)
    buf = DeviceBuffer.from_host(ctx, a)
    _lib.call("gl_coset_ntt_batch", buf.ptr, a.shape[0], log_n, a.shape[1], shift, 1, ctx.ptr)
    out = buf.download().reshape(a.shape)
    buf.free()
    return out if np.ndim(values) > 1 else out[0]
