"""Child process of test_reference_quotient.py::test_compiled_gates_equal_the_interpreter_where_the_short_forms_take_their_rare_paths: the
DIAGNOSTIC build of the library (csrc/knobs.h) reads the gate-kernel generator's switches from the environment (PLONKY2_HIP_JIT_FUSE,
_PEEPHOLE ...); the ed25519 table compiled under them must give what the interpreter gives on the same leaves of edge values. Prints the
sha256 of the result (the parent holds it against the product library's) and the generated source's marks.
usage: python tests/gate_jit_variant_child.py <edges | edges among random | non-canonical>"""
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import plonky2_gpu_amd as pg  # noqa: E402
from plonky2_gpu_amd import ed25519_circuit as ed, gate_program as gp  # noqa: E402

import test_reference_quotient as trq  # noqa: E402


def main():
    which = sys.argv[1]
    gpu = pg.Context(0)
    inst = trq.edge_instance(which)
    up = lambda a: pg.DeviceBuffer.from_host(gpu, __import__("numpy").ascontiguousarray(a).reshape(-1))  # noqa: E731
    bufs = {k: up(inst[k]) for k in ("wires", "zs", "cs", "k_is")}
    pool = gp.ImmediatePool()
    prog = pg.GateProgram(gpu, [gp.build_gate(k, p, pool) for k, p in ed.GATES], ed.SELECTOR_INDICES, ed.GROUPS,
                          ed.REFERENCE_PUBLIC_INPUTS_HASH, immediates=pool.values)
    want = trq._generic(gpu, inst, bufs, trq.EDGE_LOG_LEN, prog=prog)
    prog.compile(ed.NUM_GATE_CONSTRAINTS, 2)
    got = trq._generic(gpu, inst, bufs, trq.EDGE_LOG_LEN, kernel=prog.kernel)
    assert (got == want).all(), "the compiled kernel differs from the interpreter"
    src = prog.kernel_source()
    print("sha256", hashlib.sha256(got.tobytes()).hexdigest())
    print("marks", src.count("gl::mul_add_small<1>("), src.count("gl::sub_small<3u>("), int("g_bias[c * NGU + " in src), int("GateSum gate_8()" in src),
          int("// gate_8" in src))


if __name__ == "__main__":
    main()
