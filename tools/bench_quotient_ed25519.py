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

GATES = [("noop", None), ("constant", 2), ("public_input", None), ("base_sum", (2, 32)), ("base_sum", (2, 63)), ("arithmetic", 20),
         ("base_sum", (4, 16)), ("comparison", (32, 16)),
         ("u32_add_many", (0, 11)), ("u32_add_many", (11, 5)), ("u32_add_many", (13, 5)), ("u32_add_many", (15, 4)),
         ("u32_add_many", (16, 4)), ("u32_add_many", (2, 10)), ("u32_add_many", (3, 9)), ("u32_add_many", (5, 9)),
         ("u32_add_many", (7, 8)), ("u32_add_many", (9, 6)),
         ("u32_arithmetic", 6), ("u32_range_check", 0), ("u32_range_check", 1), ("u32_range_check", 8), ("u32_subtraction", 11),
         ("random_access", (4, 4, 2)), ("poseidon", None)]
GROUPS = [(0, 6), (6, 11), (11, 16), (16, 21), (21, 24), (24, 25)]
SELECTOR_INDICES = [0] * 6 + [1] * 5 + [2] * 5 + [3] * 5 + [4] * 3 + [5]
NUM_GATE_CONSTRAINTS = 231


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
    wires = pg.PolynomialBatch.from_values(ctx, rand_cols(rng, 234, n), 3, False, 4, leaf_major=False)
    cs = pg.PolynomialBatch.from_values(ctx, rand_cols(rng, 88, n), 3, False, 4, leaf_major=False)
    zs = pg.PolynomialBatch.from_values(ctx, rand_cols(rng, 20, n), 3, False, 4, leaf_major=False)
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
    out = dict(workload=f"compute_quotient_polys, ed25519 gate list (25 gates, {n_instr} program instructions, 231 constraints), n=2^{db}, "
                        f"234 wires / 80 routed / 8 constants, LDE 2^{db + 3}, random data",
               gate_program_instructions=n_instr, immediates=len(pool.values), hiprtc_compile_s=round(compile_s, 1),
               kernel_source_bytes=len(prog.kernel_source()), compiled_ms=round(min(times[1:]), 3))
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
