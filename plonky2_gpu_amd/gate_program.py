"""Host-side assembler for the device's table-driven gate-constraint evaluator.

The device evaluates `evaluate_gate_constraints_base_batch` (plonky2/src/plonk/vanishing_poly.rs:267-306)
for any circuit from a small register program per gate — what a Rust host would emit from each
gate's `eval_unfiltered_*` (the same way plonky2 already derives the recursive-circuit evaluator).
Instruction = 4 x u16 {op, dst, a, b}; registers hold field elements.

    LOAD_WIRE  dst <- local_wires[a]
    LOAD_CONST dst <- local_constants[num_selectors + a]        (vars.remove_prefix, gates/gate.rs:104)
    LOAD_PI    dst <- public_inputs_hash[a]
    LOAD_IMM   dst <- immediates[a]
    ADD/SUB/MUL dst <- r[a] (op) r[b]
    EMIT       next constraint of this gate <- r[a]            (accumulated as filter * r[a])
"""
import numpy as np

LOAD_WIRE, LOAD_CONST, LOAD_PI, LOAD_IMM, ADD, SUB, MUL, EMIT = range(8)
MAX_REGS = 64


class GateAsm:
    def __init__(self):
        self.instrs = []
        self.next_reg = 0

    def _reg(self):
        r = self.next_reg
        self.next_reg += 1
        if r >= MAX_REGS:
            raise ValueError("gate program needs more than %d registers" % MAX_REGS)
        return r

    def op(self, code, a, b=0):
        r = self._reg()
        self.instrs.append((code, r, a, b))
        return r

    def wire(self, i):
        return self.op(LOAD_WIRE, i)

    def const(self, i):
        return self.op(LOAD_CONST, i)

    def pi(self, i):
        return self.op(LOAD_PI, i)

    def add(self, a, b):
        return self.op(ADD, a, b)

    def sub(self, a, b):
        return self.op(SUB, a, b)

    def mul(self, a, b):
        return self.op(MUL, a, b)

    def emit(self, a):
        self.instrs.append((EMIT, 0, a, 0))

    def release(self):
        """registers are per constraint group: callers may reset between independent constraints"""
        self.next_reg = 0


def arithmetic_gate(num_ops):
    """ArithmeticGate { num_ops } (plonky2/src/gates/arithmetic_base.rs:199-216)"""
    g = GateAsm()
    for i in range(num_ops):
        g.release()
        c0, c1 = g.const(0), g.const(1)
        m0, m1, ad, out = g.wire(4 * i), g.wire(4 * i + 1), g.wire(4 * i + 2), g.wire(4 * i + 3)
        computed = g.add(g.mul(g.mul(m0, m1), c0), g.mul(ad, c1))
        g.emit(g.sub(out, computed))
    return g.instrs


def constant_gate(num_consts):
    """ConstantGate { num_consts } (plonky2/src/gates/constant.rs:150-158)"""
    g = GateAsm()
    for i in range(num_consts):
        g.release()
        g.emit(g.sub(g.const(i), g.wire(i)))
    return g.instrs


def public_input_gate():
    """PublicInputGate (plonky2/src/gates/public_input.rs:129-139)"""
    g = GateAsm()
    for i in range(4):
        g.release()
        g.emit(g.sub(g.wire(i), g.pi(i)))
    return g.instrs


def noop_gate():
    return []


def pack_program(gate_instrs, selector_indices, groups):
    """-> (instrs u16[n,4], gate descriptors u32[g,6]) for GlGateProgram."""
    instrs, descs = [], []
    for row, ins in enumerate(gate_instrs):
        si = selector_indices[row]
        descs.append((row, si, groups[si][0], groups[si][1], len(instrs), len(ins)))
        instrs += ins
    a = np.array(instrs if instrs else [(0, 0, 0, 0)], dtype=np.uint16).reshape(-1, 4)
    d = np.array(descs, dtype=np.uint32).reshape(-1, 6)
    return a, d
