"""The gate emitters (plonky2_gpu_amd/gate_program.py) checked on the CPU: every register program of the ed25519
gate table is executed with Python integers — a direct reading of the opcode definitions in include/plonky2_hip.h,
including the two halves of an ACC accumulator and the bound both must respect — on random rows, and its
constraints are compared with the oracle's restatement of the same gate (oracle/gates_ref.py, oracle/plonk_ref.py).
The GPU suite runs the same programs through the interpreter and the run-time compiled kernel."""
import random

import pytest

from oracle import prove_ref
from plonky2_gpu_amd import ed25519_circuit as ed
from plonky2_gpu_amd import gate_program as gp

P = 0xFFFFFFFF00000001


def execute(instrs, imms, num_selectors, consts, wires, pih):
    """one gate's program on one row -> the list of emitted constraint values (mod p)"""
    regs, out = {}, []
    acc_lo, acc_hi = [0] * 4, [0] * 4
    for op, dst, a, b in instrs:
        if op == gp.LOAD_WIRE:
            regs[dst] = wires[a]
        elif op == gp.LOAD_CONST:
            regs[dst] = consts[num_selectors + a]
        elif op == gp.LOAD_PI:
            regs[dst] = pih[a]
        elif op == gp.LOAD_IMM:
            regs[dst] = imms[a] % P
        elif op == gp.ADD:
            regs[dst] = (regs[a] + regs[b]) % P
        elif op == gp.SUB:
            regs[dst] = (regs[a] - regs[b]) % P
        elif op == gp.MUL:
            regs[dst] = regs[a] * regs[b] % P
        elif op == gp.MULK:
            regs[dst] = (regs[a] << b) % P
        elif op == gp.EMIT:
            out.append(regs[a])
        elif op == gp.ACC:
            # registers hold ANY u64 representative on the device: take the worst case for the bound, 2^32 - 1 per half
            assert imms[b] < 1 << 32
            x = regs[a]
            acc_lo[dst] += (x & 0xFFFFFFFF) * imms[b]
            acc_hi[dst] += (x >> 32) * imms[b]
            execute.worst[dst] += 0xFFFFFFFF * imms[b]
            assert execute.worst[dst] < 1 << 63, "an accumulator half could wrap"
        elif op == gp.ACCR:
            regs[dst] = (acc_lo[a] + (acc_hi[a] << 32)) % P
            acc_lo[a] = acc_hi[a] = 0
            execute.worst[a] = 0
        else:
            raise AssertionError("unknown opcode %d" % op)
    assert acc_lo == [0] * 4 and acc_hi == [0] * 4, "an accumulator was left unreduced"
    return out


@pytest.mark.parametrize("row", range(len(ed.GATES)))
def test_emitted_program_equals_the_oracle_gate(row):
    kind, param = ed.GATES[row]
    pool = gp.ImmediatePool()
    instrs = gp.build_gate(kind, param, pool)
    oracle_gate = prove_ref.base_gates({"gates": [(kind, param)]})[0]
    rng = random.Random(1000 + row)
    num_selectors = len(ed.GROUPS)
    for trial in range(3):
        # trial 0: small values (limb-like), trial 1: random field elements, trial 2: the top of the field
        draw = [lambda: rng.randrange(4), lambda: rng.randrange(P), lambda: P - 1 - rng.randrange(3)][trial]
        wires = [draw() for _ in range(ed.NUM_WIRES)]
        consts = [rng.randrange(P) for _ in range(ed.NUM_CONSTANTS)]
        pih = [rng.randrange(P) for _ in range(4)]
        execute.worst = [0] * 4
        got = execute(instrs, pool.values, num_selectors, consts, wires, pih)
        exp = [int(v) % P for v in oracle_gate(consts[num_selectors:], wires, pih)]
        assert got == exp, (kind, param, trial)


UPSTREAM_GATES = [("arithmetic_extension", 10), ("mul_extension", 13), ("reducing", 43), ("reducing_extension", 32), ("exponentiation", 66),
                  ("exponentiation", 1), ("poseidon_mds", None), ("low_degree_interpolation", 4), ("low_degree_interpolation", 2),
                  ("low_degree_interpolation", 1), ("high_degree_interpolation", 1), ("high_degree_interpolation", 2), ("high_degree_interpolation", 3),
                  ("reducing", 1), ("reducing_extension", 1)]


@pytest.mark.parametrize("kind,param", UPSTREAM_GATES)
def test_upstream_gate_programs_equal_the_oracle_gates(kind, param):
    """The eight gate kinds of upstream plonky2 beyond the ed25519 list (round 5; standard_recursion_config's parameters and small
    ones): the emitted program, executed with Python integers, gives oracle/gates_ref.py's constraints on small, random and
    top-of-the-field rows, and zeros on an honestly generated row."""
    from oracle import gates_ref

    pool = gp.ImmediatePool()
    instrs = gp.build_gate(kind, param, pool)
    rng = random.Random(hash((kind, str(param))) & 0xFFFF)
    width = gates_ref.num_wires(kind, param)
    for trial in range(4):
        consts = [rng.randrange(P) for _ in range(6)]
        pih = [rng.randrange(P) for _ in range(4)]
        if trial == 3:
            wires = gates_ref.fill_row(kind, param, rng, consts[4:], pih)
        else:
            draw = [lambda: rng.randrange(4), lambda: rng.randrange(P), lambda: P - 1 - rng.randrange(3)][trial]
            wires = [draw() for _ in range(width)]
        execute.worst = [0] * 4
        got = execute(instrs, pool.values, 4, consts, wires + [0] * 4, pih)
        exp = gates_ref.constraints(kind, param, consts[4:], wires + [0] * 4, pih, gates_ref.Base)
        assert got == exp, (kind, param, trial)
        assert len(got) == gates_ref.num_constraints(kind, param)
        if trial == 3:
            assert not any(got)


def test_accumulator_bound_is_enforced_by_the_emitter():
    g = gp.GateAsm(gp.ImmediatePool())
    x = g.wire(0)
    g.acc(x, (1 << 31) - 1)
    with pytest.raises(ValueError):
        g.acc(x, 1 << 32)  # weight does not fit 32 bits
    with pytest.raises(ValueError):
        g.acc(x, 1 << 31)  # (2^31 - 1 + 2^31) * (2^32 - 1) >= 2^63
    g.accr()
    g.acc(x, 1 << 31)  # fine again after the reduction... but only just below the limit: nothing more fits
    assert not g.acc_fits(1 << 20)
    # reduce_with_powers splits by itself: 63 bits need three blocks, 16 base-4 limbs need one
    for base, count, blocks in ((2, 63, 3), (4, 16, 1), (4, 17, 2), (16, 16, 2)):
        g = gp.GateAsm(gp.ImmediatePool())
        g.reduce_with_powers(list(range(count)), base, wires=True)
        assert sum(1 for i in g.instrs if i[0] == gp.ACCR) == blocks, (base, count)
