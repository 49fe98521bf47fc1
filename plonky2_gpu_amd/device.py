"""Device context and buffers on top of the C ABI (numpy on the host side, no torch needed)."""
import ctypes

import numpy as np

from . import _lib


class Context:
    """Two HIP streams on one device: the twin of the reference's CudaInnerContext
    (plonky2/src/fri/oracle.rs:43-47)."""

    def __init__(self, device=0):
        lib = _lib.load()
        if lib.gl_device_count() <= 0:
            raise RuntimeError("plonky2_gpu_amd: no HIP device visible (there is no CPU fallback)")
        self.device = device
        self.ptr = lib.gl_ctx_create(device)
        if not self.ptr:
            raise RuntimeError(f"gl_ctx_create({device}) failed")

    def synchronize(self):
        _lib.call("gl_ctx_synchronize", self.ptr)

    def close(self):
        if self.ptr:
            _lib.load().gl_ctx_destroy(self.ptr)
            self.ptr = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


class DeviceBuffer:
    """n_elems u64 field elements in HBM, on the device of `ctx` (hipMalloc through gl_ctx_malloc)."""

    def __init__(self, ctx, n_elems):
        self.ctx = ctx
        self.n = int(n_elems)
        p = ctypes.c_void_p()
        _lib.call("gl_ctx_malloc", ctypes.byref(p), self.n * 8, ctx.ptr)
        self.ptr = p.value

    @classmethod
    def from_host(cls, ctx, arr):
        a = np.ascontiguousarray(arr, dtype=np.uint64)
        buf = cls(ctx, a.size)
        buf.upload(a)
        return buf

    def upload(self, arr, offset=0):
        a = np.ascontiguousarray(arr, dtype=np.uint64)
        assert offset + a.size <= self.n
        if a.size:
            _lib.call("gl_memcpy_h2d", self.ptr + offset * 8, a.ctypes.data, a.size * 8, self.ctx.ptr)

    def upload_async(self, pinned, offset=0):
        """queue the copy of a PinnedArray (or a numpy array the caller keeps alive and unchanged) on the context's
        second stream and return at once; ctx.synchronize() before the data is used"""
        a = pinned.array if isinstance(pinned, PinnedArray) else pinned
        assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"] and offset + a.size <= self.n
        if a.size:
            _lib.call("gl_memcpy_h2d_async", self.ptr + offset * 8, a.ctypes.data, a.size * 8, self.ctx.ptr)

    def download(self, offset=0, count=None):
        count = self.n - offset if count is None else int(count)
        out = np.empty(count, dtype=np.uint64)
        if count:
            _lib.call("gl_memcpy_d2h", out.ctypes.data, self.ptr + offset * 8, count * 8, self.ctx.ptr)
        return out

    def at(self, offset):
        return self.ptr + int(offset) * 8

    def free(self):
        if self.ptr:
            _lib.call("gl_free", self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class PinnedArray:
    """A page-locked host array of u64 (hipHostMalloc): the twin of the reference's pinned staging
    vectors (MyAllocator, plonky2/src/fri/oracle.rs:49-73). `.array` is a numpy view."""

    def __init__(self, n_elems):
        p = ctypes.c_void_p()
        _lib.call("gl_malloc_host", ctypes.byref(p), int(n_elems) * 8)
        self.ptr = p.value
        self.array = np.ctypeslib.as_array(ctypes.cast(self.ptr, ctypes.POINTER(ctypes.c_uint64)), shape=(int(n_elems),))

    def free(self):
        if self.ptr:
            self.array = None
            _lib.call("gl_free_host", self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Event:
    def __init__(self):
        p = ctypes.c_void_p()
        _lib.call("gl_event_create", ctypes.byref(p))
        self.ptr = p.value

    def record(self, ctx):
        _lib.call("gl_event_record", self.ptr, ctx.ptr)

    def elapsed_ms_since(self, start):
        ms = ctypes.c_float()
        _lib.call("gl_event_elapsed_ms", ctypes.byref(ms), start.ptr, self.ptr)
        return ms.value

    def __del__(self):
        try:
            if self.ptr:
                _lib.load().gl_event_destroy(self.ptr)
        except Exception:
            pass
