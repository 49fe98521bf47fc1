"""gl_gate_kernel_build cuts a circuit's gates into units that hiprtc compiles side by side (csrc/gate_jit.hip): hiprtc needs no
GPU, so the cutting, the cache files and the error path are checked here; without a device the build stops at loading the code
objects, after they have been written. tests/test_gpu_plonk.py compares the kernels' results with the interpreter and the oracle."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import ctypes, os, sys, time
sys.path.insert(0, {root!r})
import numpy as np
from plonky2_gpu_amd import _lib, gate_program as gp
pool = gp.ImmediatePool()
kinds = [("arithmetic", 20), ("constant", 2), ("public_input", None), ("noop", None), ("base_sum", (2, 63)), ("u32_range_check", 8), ("comparison", (32, 16))]
programs = [gp.build_gate(k, p, pool) for k, p in kinds]
instrs = np.concatenate([np.asarray(p, dtype=np.uint16).reshape(-1, 4) for p in programs if len(p)] or [np.zeros((0, 4), np.uint16)]).reshape(-1)
descs, pc = [], 0
for g, p in enumerate(programs):
    descs += [g, 0, 0, len(programs), pc, len(p)]
    pc += len(p)
descs = np.asarray(descs, dtype=np.uint32)
imms = np.asarray(pool.values, dtype=np.uint64)
ngc = max(1, max(sum(1 for ins in p if ins[0] == gp.EMIT) for p in programs))
wires = 1 + max([ins[2] for p in programs for ins in p if ins[0] == gp.LOAD_WIRE] or [0])
if os.environ.get("BREAK"):
    instrs = instrs.copy(); instrs[0] = 99  # unknown opcode
k = ctypes.c_void_p()
t = time.perf_counter()
try:
    _lib.call("gl_gate_kernel_build", instrs, instrs.size // 4, descs, descs.size // 6, imms, imms.size, 1, ngc, 2, ctypes.byref(k))
    print("built")
except _lib.Plonky2HipError as e:
    print("error:", str(e)[:300].replace("\n", " "))
print("seconds %.2f" % (time.perf_counter() - t))
"""


DEBUG_LIB = os.path.join(ROOT, "plonky2_gpu_amd", "libplonky2_hip_debug.so")


def run(tmp, **env):
    e = dict(os.environ, PLONKY2_HIP_KERNEL_CACHE=str(tmp), AMD_COMGR_CACHE="0", **env)
    if any(k.startswith("PLONKY2_HIP_JIT_") and k != "PLONKY2_HIP_JIT_FORK" for k in env):
        # the generator's switches are read by the diagnostic build only (csrc/knobs.h)
        assert os.path.exists(DEBUG_LIB), "make -C plonky2_gpu_amd/csrc debug (done by __graft_entry__.build())"
        e["PLONKY2_HIP_LIBRARY"] = DEBUG_LIB
    r = subprocess.run([sys.executable, "-c", CHILD.format(root=ROOT)], env=e, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    return r.stdout


def objects(tmp):
    return sorted(f for f in os.listdir(tmp) if f.endswith(".hsaco"))


def test_fused_units_are_compiled_into_the_cache_and_forking_changes_nothing(tmp_path):
    """The default generator: gates that share values share a unit (csrc/gate_jit.hip, fused units). Of the seven gates here the noop
    gate has no constraints and belongs to no unit; at most PLONKY2_HIP_JIT_FUSE_GATES gates go into one."""
    a, b, c = tmp_path / "a", tmp_path / "b", tmp_path / "c"
    for d in (a, b, c):
        d.mkdir()
    out = run(a, PLONKY2_HIP_JIT_FORK="1")
    assert "built" in out or "loading the compiled gate kernel" in out, out
    n = len(objects(a))
    assert 2 <= n <= 6 and len([f for f in os.listdir(a) if f.endswith(".hip")]) == n, os.listdir(a)
    sources = "".join((a / f).read_text() for f in os.listdir(a) if f.endswith(".hip"))
    assert all(("// gate_%d\n" % g) in sources for g in (0, 1, 2, 4, 5, 6)) and "// gate_3\n" not in sources  # gate 3 is the noop gate
    assert "GateSum gate_" not in sources
    run(b)  # no fork: same sources -> same names, same code objects
    assert objects(a) == objects(b)
    for f in objects(a):
        assert (a / f).read_bytes() == (b / f).read_bytes(), f
    run(c, PLONKY2_HIP_JIT_FUSE_GATES="1")  # one gate per unit (a switch of the diagnostic build)
    assert len(objects(c)) == 6
    d = tmp_path / "d"
    d.mkdir()
    e = dict(os.environ, PLONKY2_HIP_KERNEL_CACHE=str(d), PLONKY2_HIP_JIT_FUSE_GATES="1")  # the PRODUCT library does not read it
    e.pop("PLONKY2_HIP_LIBRARY", None)
    r = subprocess.run([sys.executable, "-c", CHILD.format(root=ROOT)], env=e, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and objects(d) == objects(a), r.stdout[-500:] + r.stderr[-1000:]
    out = run(a)
    assert float(out.split("seconds")[1]) < 2.0, out


def test_units_are_compiled_into_the_cache_and_forking_changes_nothing(tmp_path, monkeypatch):
    """One function per gate (PLONKY2_HIP_JIT_FUSE=0, the generator of rounds 3-4, kept for A/B in the diagnostic build):
    PLONKY2_HIP_JIT_UNITS units."""
    monkeypatch.setenv("PLONKY2_HIP_JIT_FUSE", "0")
    monkeypatch.setenv("PLONKY2_HIP_LIBRARY", DEBUG_LIB)
    assert os.path.exists(DEBUG_LIB), "make -C plonky2_gpu_amd/csrc debug (done by __graft_entry__.build())"
    a, b, c = tmp_path / "a", tmp_path / "b", tmp_path / "c"
    for d in (a, b, c):
        d.mkdir()
    out = run(a, PLONKY2_HIP_JIT_UNITS="3", PLONKY2_HIP_JIT_FORK="1")  # forked children: opt-in (what build() uses)
    assert "built" in out or "loading the compiled gate kernel" in out, out  # no device here: the build ends at the module load
    assert len(objects(a)) == 3 and len([f for f in os.listdir(a) if f.endswith(".hip")]) == 3, os.listdir(a)
    assert not [f for f in os.listdir(a) if ".tmp." in f or f.count(".hip.")], os.listdir(a)  # nothing half-written left behind
    # the same units from this process alone, one after the other: same sources -> same names, same code objects
    run(b, PLONKY2_HIP_JIT_UNITS="3")  # the default: no fork
    assert objects(a) == objects(b)
    for f in objects(a):
        assert (a / f).read_bytes() == (b / f).read_bytes(), f
    # one unit: one program with every gate in it
    run(c, PLONKY2_HIP_JIT_UNITS="1")
    assert len(objects(c)) == 1
    # a second build finds everything in the cache (no compilation: well under a second of hiprtc)
    out = run(a, PLONKY2_HIP_JIT_UNITS="3")
    assert float(out.split("seconds")[1]) < 2.0, out


def test_a_program_that_cannot_be_generated_is_reported(tmp_path):
    out = run(tmp_path, BREAK="1")
    assert "error:" in out and "opcode" in out, out
    assert objects(tmp_path) == []


def test_build_leaves_exactly_the_current_units_in_the_kernel_cache(tmp_path):
    """__graft_entry__.precompile_gate_kernels() (part of build()): whatever the cache held — units of an earlier generator, a unit
    missing — afterwards it holds the units of the current generator for the compiled-in ed25519 table, all of them and nothing else,
    and a unit that is already there is not compiled again. Run in a FRESH process on a COPY of the shipped cache: the function forks
    hiprtc workers, which a process that has initialised HIP (this pytest process, on a GPU box) must not do, and the shipped cache
    is a build product that a test has no business rewriting."""
    import shutil
    import subprocess

    shipped = os.path.join(ROOT, "plonky2_gpu_amd", "kernel_cache")
    cache = str(tmp_path / "kernel_cache")
    if os.path.isdir(shipped):
        shutil.copytree(shipped, cache)

    def precompile():
        env = {k: v for k, v in os.environ.items() if not k.startswith("PLONKY2_HIP_")}
        r = subprocess.run([sys.executable, "-c", "import sys, os, __graft_entry__ as g; g.precompile_gate_kernels(cache=sys.argv[1]); "
                            "assert 'PLONKY2_HIP_JIT_FORK' not in os.environ and 'PLONKY2_HIP_KERNEL_CACHE' not in os.environ", cache],
                           cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
        assert r.returncode == 0, r.stdout + r.stderr
        return r.stdout

    precompile()
    good = set(os.listdir(cache))
    assert good and len([f for f in good if f.endswith(".hsaco")]) * 2 == len(good), good  # a .hip beside every .hsaco
    stamp = {f: os.stat(os.path.join(cache, f)).st_mtime_ns for f in good}
    victim = sorted(f for f in good if f.endswith(".hsaco"))[0]
    os.remove(os.path.join(cache, victim))
    os.remove(os.path.join(cache, victim[:-6] + ".hip"))
    for stale in ("gate_0000000000000000.hsaco", "gate_0000000000000000.hip"):
        with open(os.path.join(cache, stale), "wb") as f:
            f.write(b"stale")
    out = precompile()
    assert set(os.listdir(cache)) == good and victim in out, out
    for f in good - {victim, victim[:-6] + ".hip"}:  # the units that were there were used as they were
        assert os.stat(os.path.join(cache, f)).st_mtime_ns == stamp[f], f
    assert "already compiled" in precompile()  # and a third call changes nothing
    assert set(os.listdir(cache)) == good
