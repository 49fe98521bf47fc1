"""compute_quotient_polys with the REAL gate list of the plonky2-ed25519 circuit (SURVEY.md Appendix B:
25 gates, 6 selector groups, 231 gate constraints, 234 wires) at the real shape (n = 2^18, LDE 2^21) on
random data — the stage's cost does not depend on the witness being satisfying. The gates arrive as
register programs and run through the run-time compiled kernel (and once through the interpreter).
usage: python tools/bench_quotient_ed25519.py [degree_bits=18] [reps=3] [interpreter=1]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np  # noqa: E402

import plonky2_gpu_amd as pg  # noqa: E402
from plonky2_gpu_amd import gate_program as gp  # noqa: E402

from plonky2_gpu_amd.ed25519_circuit import GATES, GROUPS, NUM_GATE_CONSTRAINTS, SELECTOR_INDICES  # noqa: E402


def rand_cols(rng, cols, n):
    return rng.integers(0, pg.P, size=(cols, n), dtype=np.uint64)


def main():
    db = int(sys.argv[1]) if len(sys.argv) > 1 else 18
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    interp = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    ctx = pg.Context(0)
    n = 1 << db
    rng = np.random.default_rng(3)
    pool = gp.ImmediatePool()
    programs = [gp.build_gate(k, p, pool) for k, p in GATES]
    n_instr = sum(len(p) for p in programs)
    prog = pg.GateProgram(ctx, programs, SELECTOR_INDICES, GROUPS, [1, 2, 3, 4], immediates=pool.values)
    t = time.perf_counter()
    prog.compile(NUM_GATE_CONSTRAINTS, 2)
    compile_s = time.perf_counter() - t
    # leaf_major=True: the commitments also keep the leaf-major copy, the layout the reference's symbol reads
    wires = pg.PolynomialBatch.from_values(ctx, rand_cols(rng, 234, n), 3, False, 4, leaf_major=True)
    cs = pg.PolynomialBatch.from_values(ctx, rand_cols(rng, 88, n), 3, False, 4, leaf_major=True)
    zs = pg.PolynomialBatch.from_values(ctx, rand_cols(rng, 20, n), 3, False, 4, leaf_major=True)
    d_k = pg.DeviceBuffer.from_host(ctx, np.array([pow(7, j, pg.P) for j in range(80)], dtype=np.uint64))
    ch = [int(x) for x in rng.integers(1, pg.P, size=6, dtype=np.uint64)]

    def run(p):
        ctx.synchronize()
        t0 = time.perf_counter()
        d = pg.compute_quotient_polys(ctx, wires, cs, zs, 8, 80, d_k, ch[0:2], ch[2:4], ch[4:6], 8, None, NUM_GATE_CONSTRAINTS, p)
        ctx.synchronize()
        return (time.perf_counter() - t0) * 1e3, d

    times = []
    for _ in range(reps + 1):
        ms, d_jit = run(prog)
        times.append(ms)
    # the same call without any gate constraints: the permutation terms, Z_H, the Horner sums and the two inverse transforms
    def run_without_gates():
        ctx.synchronize()
        t0 = time.perf_counter()
        d = pg.compute_quotient_polys(ctx, wires, cs, zs, 8, 80, d_k, ch[0:2], ch[2:4], ch[4:6], 8, None, 0, None)
        ctx.synchronize()
        d.free()
        return (time.perf_counter() - t0) * 1e3

    without = min(run_without_gates() for _ in range(reps + 1))
    out = dict(workload=f"compute_quotient_polys, ed25519 gate list (25 gates, {n_instr} program instructions, 231 constraints), n=2^{db}, "
                        f"234 wires / 80 routed / 8 constants, LDE 2^{db + 3}, random data",
               gate_program_instructions=n_instr, immediates=len(pool.values), hiprtc_compile_s=round(compile_s, 1),
               kernel_source_bytes=len(prog.kernel_source()), compiled_ms=round(min(times[1:]), 3),
               without_gate_constraints_ms=round(without, 3))
    # the reference's own symbol `compute_quotient_polys` on the same data: circuit compiled into the library,
    # LEAF-MAJOR reads (its contract), challenges in device memory
    up = lambda v: pg.DeviceBuffer.from_host(ctx, np.array(v, dtype=np.uint64))  # noqa: E731
    d_be, d_ga, d_al = up(ch[0:2]), up(ch[2:4]), up(ch[4:6])
    pg.reference_set_public_inputs_hash([1, 2, 3, 4])
    sym = []
    for _ in range(reps + 1):
        ctx.synchronize()
        t0 = time.perf_counter()
        d_sym = pg.reference_compute_quotient_polys(ctx, wires.merkle_tree.d_leaves, db, zs.merkle_tree.d_leaves, cs.merkle_tree.d_leaves,
                                                    d_k, d_al, d_be, d_ga)
        ctx.synchronize()
        sym.append((time.perf_counter() - t0) * 1e3)
    out["reference_symbol_ms"] = round(min(sym[1:]), 3)  # transposes into the library's column-major staging buffer first
    out["reference_symbol_first_call_ms"] = round(sym[0], 1)
    out["reference_symbol_equals_generic"] = bool((d_sym.download() == d_jit.download()).all())
    os.environ["PLONKY2_HIP_REFERENCE_IN_PLACE"] = "1"  # the same call reading the leaf-major rows in place
    sym = []
    for _ in range(reps):
        ctx.synchronize()
        t0 = time.perf_counter()
        d_inp = pg.reference_compute_quotient_polys(ctx, wires.merkle_tree.d_leaves, db, zs.merkle_tree.d_leaves, cs.merkle_tree.d_leaves,
                                                    d_k, d_al, d_be, d_ga)
        ctx.synchronize()
        sym.append((time.perf_counter() - t0) * 1e3)
    del os.environ["PLONKY2_HIP_REFERENCE_IN_PLACE"]
    pg.reference_set_public_inputs_hash(None)
    out["reference_symbol_in_place_ms"] = round(min(sym), 3)
    out["reference_symbol_in_place_equals_staged"] = bool((d_inp.download() == d_sym.download()).all())
    if interp:
        kernel, prog.kernel = prog.kernel, None  # same programs through the interpreter
        ms, d_int = run(prog)
        ms, d_int = run(prog)
        prog.kernel = kernel
        out["interpreter_ms"] = round(ms, 3)
        out["interpreter_equals_compiled"] = bool((d_int.download() == d_jit.download()).all())
    print(json.dumps(out))


if __name__ == "__main__":
    main()
