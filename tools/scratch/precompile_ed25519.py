"""Compile the ed25519 gate-list kernel WITHOUT a GPU into the on-disk cache (the build writes the code
object before it tries to load it), so that GPU runs start from a warm $PLONKY2_HIP_KERNEL_CACHE.
(A cold hiprtc build of this kernel takes about a minute; ~2 s when comgr's own cache has the source.)"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np  # noqa: E402

import gen_ed25519_program as g  # noqa: E402
from plonky2_gpu_amd import _lib  # noqa: E402

instrs, descs, imms = g.arrays()
imms = np.array(imms, dtype=np.uint64)
k = ctypes.c_void_p()
try:
    _lib.call("gl_gate_kernel_build", instrs, instrs.size // 4, descs, descs.size // 6, imms, imms.size, 6, 231, 2, ctypes.byref(k))
except Exception as e:
    print("(expected without a device)", str(e)[:90])
