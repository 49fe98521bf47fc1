"""gl_gate_programs_emit (csrc/gate_emit.hip): the native emitters of the gate register programs give, instruction for
instruction and immediate for immediate, what plonky2_gpu_amd/gate_program.py gives — for every gate kind over a range of
parameters, and for the whole ed25519 gate table. (The Python emitters are themselves checked against the oracle's gate
restatements by tests/test_gate_programs_cpu.py.) Host code only: runs without a GPU."""
import ctypes
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _emit(gates, selector_indices, groups):
    from plonky2_gpu_amd import _lib

    lib = _lib.load()
    specs = (_lib.GlGateSpec * max(1, len(gates)))()
    for i, (kind, param) in enumerate(gates):
        ps = [] if param is None else ([param] if isinstance(param, int) else list(param))
        specs[i].kind = _lib.GATE_KINDS[kind]
        for j, v in enumerate(ps):
            specs[i].params[j] = v
        specs[i].selector_index = selector_indices[i]
    bounds = np.ascontiguousarray(np.array(groups, dtype=np.uint32).reshape(-1))
    out = _lib.GlGatePrograms()
    err = lib.gl_gate_programs_emit(ctypes.byref(specs), len(gates), bounds.ctypes.data, len(groups), ctypes.byref(out))
    if err.code != 0:
        msg = ctypes.string_at(err.message).decode()
        raise RuntimeError(msg)
    instrs = np.ctypeslib.as_array(ctypes.cast(out.instrs, ctypes.POINTER(ctypes.c_uint16)), shape=(out.num_instrs, 4)).copy()
    descs = np.ctypeslib.as_array(ctypes.cast(out.gates, ctypes.POINTER(ctypes.c_uint32)), shape=(out.num_gates, 6)).copy() if out.num_gates else \
        np.zeros((0, 6), dtype=np.uint32)
    imms = np.ctypeslib.as_array(ctypes.cast(out.immediates, ctypes.POINTER(ctypes.c_uint64)), shape=(out.num_immediates,)).copy() \
        if out.num_immediates else np.zeros(0, dtype=np.uint64)
    ngc = out.num_gate_constraints
    lib.gl_gate_programs_free(ctypes.byref(out))
    assert not out.instrs and not out.gates and not out.immediates
    return instrs, descs, imms, ngc


def _python(gates, selector_indices, groups):
    from plonky2_gpu_amd import gate_program as gp

    pool = gp.ImmediatePool()
    progs = [gp.build_gate(k, p, pool) for k, p in gates]
    instrs, descs = gp.pack_program(progs, selector_indices, groups)
    ngc = max([sum(1 for ins in p if ins[0] == gp.EMIT) for p in progs] + [0])
    return instrs, descs, np.array(pool.values, dtype=np.uint64), ngc


CASES = [("noop", None), ("constant", 1), ("constant", 2), ("public_input", None), ("arithmetic", 1), ("arithmetic", 20), ("base_sum", (2, 1)),
         ("base_sum", (2, 63)), ("base_sum", (4, 16)), ("base_sum", (4, 31)), ("base_sum", (3, 7)), ("u32_add_many", (2, 3)), ("u32_add_many", (16, 4)),
         ("u32_add_many", (3, 5)), ("u32_arithmetic", 1), ("u32_arithmetic", 6), ("u32_subtraction", 1), ("u32_subtraction", 11),
         ("u32_range_check", 1), ("u32_range_check", 8), ("comparison", (32, 16)), ("comparison", (2, 1)), ("comparison", (10, 3)),
         ("random_access", (1, 2, 0)), ("random_access", (4, 4, 2)), ("random_access", (3, 1, 5)), ("poseidon", None),
         # the gates of upstream plonky2 beyond the ed25519 list, at the parameters of standard_recursion_config (135 wires, 80 routed) and small ones
         ("arithmetic_extension", 10), ("arithmetic_extension", 1), ("mul_extension", 13), ("mul_extension", 2), ("reducing", 43), ("reducing", 1),
         ("reducing_extension", 32), ("reducing_extension", 1), ("exponentiation", 66), ("exponentiation", 1), ("poseidon_mds", None),
         ("low_degree_interpolation", 4), ("low_degree_interpolation", 2), ("low_degree_interpolation", 1), ("high_degree_interpolation", 2),
         ("high_degree_interpolation", 1), ("high_degree_interpolation", 3)]


@pytest.mark.parametrize("kind,param", CASES)
def test_every_gate_kind_equals_the_python_emitter(kind, param):
    a = _emit([(kind, param)], [0], [(0, 1)])
    b = _python([(kind, param)], [0], [(0, 1)])
    assert a[3] == b[3]
    assert a[0].shape == b[0].shape and (a[0] == b[0]).all()
    assert (a[1] == b[1]).all() and (a[2] == b[2]).all()


def test_the_ed25519_gate_table_equals_the_python_emitter_and_the_compiled_in_copy():
    from plonky2_gpu_amd import ed25519_circuit as ed

    a = _emit(ed.GATES, ed.SELECTOR_INDICES, ed.GROUPS)
    b = _python(ed.GATES, ed.SELECTOR_INDICES, ed.GROUPS)
    assert a[3] == b[3] == ed.NUM_GATE_CONSTRAINTS
    assert a[0].shape == b[0].shape and (a[0] == b[0]).all() and (a[1] == b[1]).all() and (a[2] == b[2]).all()
    assert len(a[1]) == 25 and a[0].shape[0] > 20000


def test_errors_are_reported_not_crashed():
    with pytest.raises(RuntimeError, match="wider than 4 bits"):
        _emit([("comparison", (32, 4))], [0], [(0, 1)])
    with pytest.raises(RuntimeError, match="selector_index"):
        _emit([("noop", None)], [3], [(0, 1)])
    from plonky2_gpu_amd import _lib

    bad = (_lib.GlGateSpec * 1)()
    bad[0].kind = 99
    out = _lib.GlGatePrograms()
    err = _lib.load().gl_gate_programs_emit(ctypes.byref(bad), 1, np.zeros(2, dtype=np.uint32).ctypes.data, 1, ctypes.byref(out))
    assert err.code != 0 and b"no register-program emitter" in ctypes.string_at(err.message)


@pytest.mark.parametrize("kind,param,what", [
    ("arithmetic", 16384, "num_ops"), ("arithmetic", 0, "num_ops"), ("constant", 70000, "num_consts"), ("base_sum", (1, 8), "base B"),
    ("base_sum", (2, 4000), "num_limbs"), ("u32_add_many", (70000, 1), "num_addends"), ("u32_add_many", (2, 0), "num_ops"),
    ("u32_arithmetic", 100000, "num_ops"), ("u32_subtraction", 0, "num_ops"), ("u32_range_check", 5000, "num_input_limbs"),
    ("comparison", (200, 4), "num_bits"), ("comparison", (32, 0), "num_chunks"), ("random_access", (20, 1, 0), "bits"),
    ("random_access", (2, 100000, 0), "num_copies"), ("random_access", (2, 1, 100000), "num_extra_constants"),
    ("arithmetic_extension", 0, "num_ops"), ("mul_extension", 100000, "num_ops"), ("reducing", 0, "num_coeffs"), ("reducing_extension", 50000, "num_coeffs"),
    ("exponentiation", 0, "num_power_bits"), ("low_degree_interpolation", 5, "subgroup_bits"), ("high_degree_interpolation", 0, "subgroup_bits")])
def test_parameters_out_of_range_are_invalid_arguments_not_wrapped_programs(kind, param, what):
    """wire / constant indices are 16-bit instruction fields: parameters that would wrap them (or allocate without limit) come back as
    GL_E_INVALID with the parameter's name, before anything is emitted"""
    from plonky2_gpu_amd import _lib

    spec = (_lib.GlGateSpec * 1)()
    spec[0].kind = _lib.GATE_KINDS[kind]
    for j, v in enumerate([param] if isinstance(param, int) else list(param)):
        spec[0].params[j] = v
    out = _lib.GlGatePrograms()
    err = _lib.load().gl_gate_programs_emit(ctypes.byref(spec), 1, np.array([0, 1], dtype=np.uint32).ctypes.data, 1, ctypes.byref(out))
    assert err.code == -1 and what.encode() in ctypes.string_at(err.message), ctypes.string_at(err.message)
    assert not out.instrs and not out.gates and not out.immediates
