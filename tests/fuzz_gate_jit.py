"""Differential campaign for the gate-kernel generator (csrc/gate_jit.hip): random gate lists drawn from all twenty gate kinds with
random parameters, in one or two selector groups, compiled with random generator settings (gates per fused unit, statements a load is
issued ahead, waves per SIMD, fused or one function per gate, peephole pass on or off) and run through gl_compute_quotient_polys on
leaves of random field elements with edge values sprinkled in (0..5, 2^32 +- 1, p - 1.., and non-canonical representatives): the
compiled kernel must give what the INTERPRETER gives, which executes the same programs as written. Every case also checks one
setting against the default one. Not part of the test suite; run it on a GPU box:
    python tests/fuzz_gate_jit.py [cases=40] [seed=1]
Prints one line per case and a JSON summary; exits non-zero on the first mismatch."""
import ctypes
import json
import os
import random
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
# the generator's switches are read by the DIAGNOSTIC build of the library only (csrc/knobs.h)
os.environ.setdefault("PLONKY2_HIP_LIBRARY", os.path.join(ROOT, "plonky2_gpu_amd", "libplonky2_hip_debug.so"))

import plonky2_gpu_amd as pg  # noqa: E402
from plonky2_gpu_amd import _lib, gate_program as gp  # noqa: E402

P = 0xFFFFFFFF00000001
EDGES = [0, 1, 2, 3, 4, 5, (1 << 32) - 1, 1 << 32, (1 << 32) + 1, 1 << 63, P - 3, P - 2, P - 1, P, P + 1, (1 << 64) - 2, (1 << 64) - 1]
CATALOG = [("noop", None), ("constant", 2), ("public_input", None), ("arithmetic", 3), ("arithmetic", 20), ("base_sum", (2, 7)), ("base_sum", (2, 63)),
           ("base_sum", (4, 16)), ("comparison", (8, 4)), ("comparison", (32, 16)), ("u32_add_many", (0, 2)), ("u32_add_many", (2, 3)),
           ("u32_add_many", (5, 2)), ("u32_add_many", (3, 4)), ("u32_arithmetic", 2), ("u32_arithmetic", 3), ("u32_subtraction", 3),
           ("u32_subtraction", 5), ("u32_range_check", 1), ("u32_range_check", 4), ("random_access", (2, 3, 2)), ("random_access", (4, 4, 2)),
           ("arithmetic_extension", 4), ("mul_extension", 5), ("reducing", 9), ("reducing_extension", 6), ("exponentiation", 13),
           ("poseidon_mds", None), ("low_degree_interpolation", 2), ("high_degree_interpolation", 2), ("poseidon", None)]
KNOBS = ("PLONKY2_HIP_JIT_FUSE", "PLONKY2_HIP_JIT_PEEPHOLE", "PLONKY2_HIP_JIT_FUSE_GATES", "PLONKY2_HIP_JIT_PREFETCH", "PLONKY2_HIP_JIT_WAVES")


def quotient(ctx, bufs, shape, alphas, betas, gammas, prog, kernel):
    a, b, g = (np.ascontiguousarray(np.array(x, dtype=np.uint64)) for x in (alphas, betas, gammas))
    pih = np.array(prog.public_inputs_hash, dtype=np.uint64)
    n_ext = shape["n_ext"]
    work = pg.DeviceBuffer(ctx, a.size * n_ext) if kernel else None
    args = _lib.GlQuotientArgs(bufs["wires"].ptr, bufs["cs"].ptr, bufs["zs"].ptr, shape["wires"], shape["cs"], shape["zs"], bufs["k_is"].ptr, None,
                               b.ctypes.data, g.ctypes.data, a.ctypes.data, shape["num_constants"], shape["routed"], a.size, shape["ngc"],
                               shape["degree_bits"], 3, 8, 7, ctypes.pointer(prog.struct) if not kernel else None, 0, kernel,
                               pih.ctypes.data if kernel else None, work.ptr if kernel else None)
    out = pg.DeviceBuffer(ctx, a.size * n_ext)
    _lib.call("gl_compute_quotient_polys", ctypes.byref(args), out.ptr, ctx.ptr)
    res = out.download().copy()
    out.free()
    if work is not None:
        work.free()
    return res


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = random.Random(seed)
    nrng = np.random.default_rng(seed)
    ctx = pg.Context(0)
    t0 = time.time()
    for case in range(cases):
        count = rng.randint(2, 9)
        kinds = [rng.choice(CATALOG[:-1]) for _ in range(count)]
        if rng.random() < 0.15:
            kinds[rng.randrange(count)] = CATALOG[-1]  # the Poseidon gate: 4 000 operations, now and then
        two_groups = count >= 4 and rng.random() < 0.5
        cut = rng.randint(1, count - 1) if two_groups else count
        groups = [(0, cut), (cut, count)] if two_groups else [(0, count)]
        selector_indices = [0 if i < cut else 1 for i in range(count)]
        pool = gp.ImmediatePool()
        programs = [gp.build_gate(k, p, pool) for k, p in kinds]
        ngc = max(1, max(sum(1 for ins in p if ins[0] == gp.EMIT) for p in programs))
        wires = max(8, 1 + max([ins[2] for p in programs for ins in p if ins[0] == gp.LOAD_WIRE] or [0]))
        consts = 1 + max([ins[2] for p in programs for ins in p if ins[0] == gp.LOAD_CONST] or [0])
        num_constants, routed, nch = len(groups) + consts, 8, rng.choice([1, 2, 2, 3])
        degree_bits = rng.choice([2, 3, 4])
        n_ext = (1 << degree_bits) << 3
        shape = dict(n_ext=n_ext, wires=wires, cs=num_constants + routed, zs=nch, num_constants=num_constants, routed=routed, ngc=ngc, degree_bits=degree_bits)

        def leaves(width):
            x = nrng.integers(0, P, size=(n_ext, width), dtype=np.uint64)
            edge = np.array(EDGES, dtype=np.uint64)[nrng.integers(0, len(EDGES), size=x.shape)]
            return np.where(nrng.random(x.shape) < rng.choice([0.0, 0.03, 0.5]), edge, x)

        up = lambda arr: pg.DeviceBuffer.from_host(ctx, np.ascontiguousarray(arr).reshape(-1))  # noqa: E731
        bufs = dict(wires=up(leaves(wires)), cs=up(leaves(shape["cs"])), zs=up(leaves(nch)),
                    k_is=up(np.array([pow(7, j, P) for j in range(routed)], dtype=np.uint64)))
        ch = [[rng.randrange(P) for _ in range(nch)] for _ in range(3)]
        prog = pg.GateProgram(ctx, programs, selector_indices, groups, [rng.randrange(P) for _ in range(4)], immediates=pool.values)
        want = quotient(ctx, bufs, shape, *ch, prog, None)  # the interpreter
        settings = [{}]  # the default generator
        s = {"PLONKY2_HIP_JIT_FUSE_GATES": str(rng.choice([1, 2, 3, 5, 8, 16])), "PLONKY2_HIP_JIT_PREFETCH": str(rng.choice([0, 3, 16, 100])),
             "PLONKY2_HIP_JIT_WAVES": str(rng.choice([2, 3, 4]))}
        if rng.random() < 0.25:
            s = {"PLONKY2_HIP_JIT_FUSE": "0", "PLONKY2_HIP_JIT_PEEPHOLE": rng.choice(["0", "1"])}
        elif rng.random() < 0.2:
            s["PLONKY2_HIP_JIT_PEEPHOLE"] = "0"
        settings.append(s)
        ok = True
        for env in settings:
            for k in KNOBS:
                os.environ.pop(k, None)
            os.environ.update(env)
            prog.compile(ngc, nch)
            got = quotient(ctx, bufs, shape, *ch, prog, prog.kernel)
            ok &= bool((got == want).all())
        for k in KNOBS:
            os.environ.pop(k, None)
        for b in bufs.values():
            b.free()
        print(f"case {case:3d} {'ok  ' if ok else 'FAIL'} gates={[k for k, _ in kinds]} groups={groups} nch={nch} rows={n_ext} second={settings[1]}", flush=True)
        if not ok:
            print(json.dumps(dict(result="MISMATCH", case=case, seed=seed, kinds=kinds, settings=settings[1])))
            sys.exit(1)
    print(json.dumps(dict(result="all equal", cases=cases, seed=seed, seconds=round(time.time() - t0, 1))))


if __name__ == "__main__":
    main()
