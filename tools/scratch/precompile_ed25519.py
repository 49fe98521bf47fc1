"""Compile the ed25519 gate-list kernel WITHOUT a GPU into the on-disk cache (the build writes the code
object before it tries to load it), so that GPU runs of tools/bench_quotient_ed25519.py start at once."""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
from plonky2_gpu_amd import gate_program as gp, _lib
src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "bench_quotient_ed25519.py")).read()
ns = {}
exec(src[src.index("GATES = ["):src.index("def rand_cols")], ns)
pool = gp.ImmediatePool()
progs = [gp.build_gate(k, p, pool) for k, p in ns["GATES"]]
instrs, descs = gp.pack_program(progs, ns["SELECTOR_INDICES"], ns["GROUPS"])
imms = np.array(pool.values, dtype=np.uint64)
k = ctypes.c_void_p()
try:
    _lib.call("gl_gate_kernel_build", np.ascontiguousarray(instrs), instrs.size // 4, np.ascontiguousarray(descs), len(progs), imms, imms.size,
              len(ns["GROUPS"]), 231, 2, ctypes.byref(k))
except Exception as e:
    print("(expected without a device)", str(e)[:90])
