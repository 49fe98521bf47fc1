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
    MULK       dst <- r[a] * 2^b                                (b < 64)
    EMIT       next constraint of this gate <- r[a]            (accumulated as filter * r[a])
"""
import numpy as np

LOAD_WIRE, LOAD_CONST, LOAD_PI, LOAD_IMM, ADD, SUB, MUL, EMIT, MULK, ACC, ACCR = range(11)
ACC_LIMIT = 1 << 63  # each half of an accumulator stays below this (gl::fold96's precondition)
MAX_REGS = 64


class ImmediatePool:
    """Field constants referenced by LOAD_IMM, shared by all gates of a circuit (deduplicated)."""

    def __init__(self):
        self.values = []
        self._index = {}

    def index(self, value):
        value = int(value) % P
        if value not in self._index:
            if len(self.values) >= 65536:
                raise ValueError("more than 65536 distinct immediates")
            self._index[value] = len(self.values)
            self.values.append(value)
        return self._index[value]


P = 0xFFFFFFFF00000001


class GateAsm:
    """Emits one gate's register program. Registers come from a free list: `free()` returns
    temporaries, `release()` returns all of them (between independent constraints)."""

    def __init__(self, pool=None):
        self.instrs = []
        self.pool = pool
        self._free = list(range(MAX_REGS - 1, -1, -1))
        self._acc_bound = [0, 0, 0, 0]

    def _reg(self):
        if not self._free:
            raise ValueError("gate program needs more than %d live registers" % MAX_REGS)
        return self._free.pop()

    def free(self, *regs):
        for r in regs:
            if r in self._free:
                raise ValueError("register freed twice")
            self._free.append(r)

    def release(self):
        self._free = list(range(MAX_REGS - 1, -1, -1))

    def op(self, code, a, b=0, dst=None):
        r = self._reg() if dst is None else dst
        self.instrs.append((code, r, a, b))
        return r

    def wire(self, i):
        return self.op(LOAD_WIRE, i)

    def const(self, i):
        return self.op(LOAD_CONST, i)

    def pi(self, i):
        return self.op(LOAD_PI, i)

    def imm(self, value):
        if self.pool is None:
            raise ValueError("this gate needs an ImmediatePool")
        return self.op(LOAD_IMM, self.pool.index(value))

    def add(self, a, b, dst=None):
        return self.op(ADD, a, b, dst)

    def sub(self, a, b, dst=None):
        return self.op(SUB, a, b, dst)

    def mul(self, a, b, dst=None):
        return self.op(MUL, a, b, dst)

    def mulk(self, a, shift, dst=None):
        return self.op(MULK, a, shift, dst)

    def emit(self, a):
        self.instrs.append((EMIT, 0, a, 0))

    def acc_fits(self, weight, q=0):
        """may `r * weight` still be added to accumulator q without either half being able to reach 2^63?"""
        return 0 <= weight < (1 << 32) and self._acc_bound[q] + weight * 0xFFFFFFFF < ACC_LIMIT

    def acc(self, a, weight=1, q=0):
        """acc[q] += r[a] * weight, without a modular step (ACC): weight < 2^32, bound checked here"""
        if self.pool is None:
            raise ValueError("this gate needs an ImmediatePool")
        if not self.acc_fits(weight, q):
            raise ValueError("accumulator %d could overflow: reduce (accr) earlier" % q)
        self._acc_bound[q] += weight * 0xFFFFFFFF
        self.instrs.append((ACC, q, a, self.pool.index(weight)))

    def accr(self, q=0, dst=None):
        """register <- acc[q] mod p; acc[q] <- 0 (ACCR)"""
        self._acc_bound[q] = 0
        return self.op(ACCR, q, 0, dst)

    def weighted_sum(self, terms, q=0, dst=None):
        """sum_i r[reg_i] * weight_i for (reg, weight) pairs with weights < 2^32, through accumulator q; reduces
        in between (adding the partial sum back with weight 1) only if the static bound demands it"""
        started = False
        for reg, w in terms:
            if started and not self.acc_fits(w, q):
                part = self.accr(q)
                self.acc(part, 1, q)
                self.free(part)
            self.acc(reg, w, q)
            started = True
        if not started:
            return self.op(LOAD_IMM, self.pool.index(0), 0, dst)
        return self.accr(q, dst)

    # -- helpers shared by several gates ---------------------------------------------------------
    def times(self, acc, base):
        """acc <- acc * base for a small integer base: a shift when base is a power of two"""
        if base & (base - 1) == 0:
            self.mulk(acc, base.bit_length() - 1, dst=acc)
        else:
            b = self.imm(base)
            self.mul(acc, b, dst=acc)
            self.free(b)

    def reduce_with_powers(self, terms, base, wires=False, q=0):
        """sum terms[i] * base^i (plonk_common.rs:116-128) for an integer base; returns a fresh register.
        `terms` are registers, or with wires=True wire indices that are loaded for the purpose. The sum runs through
        accumulator q in blocks of consecutive terms whose weights base^j stay below 2^32 and provably fit one
        accumulation (16 base-4 limbs or 31 bits do); blocks are joined by Horner steps with base^(block length)."""
        if not terms:
            return self.imm(0)
        blocks, i = [], 0
        while i < len(terms):
            j, bound = 0, 0
            while i + j < len(terms) and base ** j < (1 << 32) and bound + base ** j * 0xFFFFFFFF < ACC_LIMIT:
                bound += base ** j * 0xFFFFFFFF
                j += 1
            for k in range(j):
                t = self.wire(terms[i + k]) if wires else terms[i + k]
                self.acc(t, base ** k, q)
                if wires:
                    self.free(t)
            blocks.append((self.accr(q), j))
            i += j
        acc = blocks[-1][0]
        for reg, length in reversed(blocks[:-1]):
            step = base ** length
            if step & (step - 1) == 0 and step.bit_length() - 1 < 64:
                self.mulk(acc, step.bit_length() - 1, dst=acc)
            else:
                m = self.imm(step % P)
                self.mul(acc, m, dst=acc)
                self.free(m)
            self.add(acc, reg, dst=acc)
            self.free(reg)
        return acc

    def range_product(self, x, small):
        """prod_{k < len(small)} (x - k); `small[k]` = register holding the constant k. The value is what
        the reference's product computes; for four factors it is formed as y (y + 2) with y = x (x - 3):
        x (x-1) (x-2) (x-3) = (x^2 - 3x) (x^2 - 3x + 2) — two multiplications instead of three."""
        if len(small) == 4:
            t = self.sub(x, small[3])
            y = self.mul(x, t)
            self.add(y, small[2], dst=t)
            self.mul(y, t, dst=y)
            self.free(t)
            return y
        acc = self.sub(x, small[0])
        for k in range(1, len(small)):
            t = self.sub(x, small[k])
            self.mul(acc, t, dst=acc)
            self.free(t)
        return acc


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


def base_sum_gate(B, num_limbs, pool):
    """BaseSumGate<B> { num_limbs } (plonky2/src/gates/base_sum.rs:213-230)"""
    g = GateAsm(pool)
    acc = g.reduce_with_powers([1 + i for i in range(num_limbs)], B, wires=True)
    s = g.wire(0)
    g.emit(g.sub(acc, s))
    g.release()
    small = [g.imm(k) for k in range(B)]
    for i in range(num_limbs):
        x = g.wire(1 + i)
        p = g.range_product(x, small)
        g.emit(p)
        g.free(x, p)
    return g.instrs


def _u32_limb_checks(g, first_limb_wire, count, split, small):
    """range-check `count` base-4 limbs (emitting one constraint each, from the LAST limb down like the
    reference's `for j in (0..n).rev()`), and return (low, high) = the limbs below / from `split`
    recombined in base 4"""
    for j in reversed(range(count)):
        limb = g.wire(first_limb_wire + j)
        p = g.range_product(limb, small)
        g.emit(p)
        if j < split:
            g.acc(limb, 4 ** j, 0)  # the two recombinations run in accumulators 0 and 1 (at most 16 limbs each fit)
        else:
            g.acc(limb, 4 ** (j - split), 1)
        g.free(limb, p)
    low = g.accr(0) if split > 0 else g.imm(0)
    high = g.accr(1) if count > split else g.imm(0)
    return low, high


def u32_add_many_gate(num_addends, num_ops, pool):
    """U32AddManyGate (u32/src/gates/add_many_u32.rs:143-184)"""
    g = GateAsm(pool)
    for i in range(num_ops):
        g.release()
        o = (num_addends + 3) * i
        small = [g.imm(k) for k in range(4)]
        if num_addends >= 3:  # addends + carry-in: plain sums, one fold
            for j in range(num_addends + 1):
                t = g.wire(o + j)
                g.acc(t, 1, 2)
                g.free(t)
            computed = g.accr(2)
        else:
            computed = g.wire(o + num_addends)
            for j in range(num_addends):
                t = g.wire(o + j)
                g.add(computed, t, dst=computed)
                g.free(t)
        res, car = g.wire(o + num_addends + 1), g.wire(o + num_addends + 2)
        comb = g.mulk(car, 32)
        g.add(comb, res, dst=comb)
        g.emit(g.sub(comb, computed, dst=comb))
        g.free(comb, computed)
        low, high = _u32_limb_checks(g, (num_addends + 3) * num_ops + 18 * i, 18, 16, small)
        g.emit(g.sub(low, res, dst=low))
        g.emit(g.sub(high, car, dst=high))
    return g.instrs


def u32_arithmetic_gate(num_ops, pool):
    """U32ArithmeticGate (u32/src/gates/arithmetic_u32.rs:326-385)"""
    g = GateAsm(pool)
    for i in range(num_ops):
        g.release()
        small = [g.imm(k) for k in range(4)]
        one, umax = small[1], g.imm(0xFFFFFFFF)
        m0, m1, ad, lo, hi, inv = (g.wire(6 * i + k) for k in range(6))
        computed = g.mul(m0, m1)
        g.add(computed, ad, dst=computed)
        t = g.sub(umax, hi)
        g.mul(inv, t, dst=t)
        g.sub(t, one, dst=t)
        g.emit(g.mul(t, lo, dst=t))
        g.mulk(hi, 32, dst=t)
        g.add(t, lo, dst=t)
        g.emit(g.sub(t, computed, dst=t))
        g.free(t, computed, m0, m1, ad, inv)
        low, high = _u32_limb_checks(g, 6 * num_ops + 32 * i, 32, 16, small)
        g.emit(g.sub(low, lo, dst=low))
        g.emit(g.sub(high, hi, dst=high))
    return g.instrs


def u32_subtraction_gate(num_ops, pool):
    """U32SubtractionGate (u32/src/gates/subtraction_u32.rs:233-269)"""
    g = GateAsm(pool)
    for i in range(num_ops):
        g.release()
        small = [g.imm(k) for k in range(4)]
        one = small[1]
        x, y, bi, res, bo = (g.wire(5 * i + k) for k in range(5))
        t = g.sub(x, y)
        g.sub(t, bi, dst=t)
        u = g.mulk(bo, 32)
        g.add(t, u, dst=t)
        g.emit(g.sub(res, t, dst=t))
        g.free(t, u, x, y, bi)
        low, _ = _u32_limb_checks(g, 5 * num_ops + 16 * i, 16, 16, small)
        g.emit(g.sub(low, res, dst=low))
        t = g.sub(one, bo)
        g.emit(g.mul(bo, t, dst=t))
    return g.instrs


def u32_range_check_gate(num_input_limbs, pool):
    """U32RangeCheckGate (u32/src/gates/range_check_u32.rs:89-111)"""
    g = GateAsm(pool)
    for i in range(num_input_limbs):
        g.release()
        small = [g.imm(k) for k in range(4)]
        aux = [g.wire(num_input_limbs + 16 * i + j) for j in range(16)]
        acc = g.reduce_with_powers(aux, 4)
        inp = g.wire(i)
        g.emit(g.sub(acc, inp, dst=acc))
        g.free(acc, inp)
        for a in aux:
            p = g.range_product(a, small)
            g.emit(p)
            g.free(p)
    return g.instrs


def comparison_gate(num_bits, num_chunks, pool):
    """ComparisonGate (u32/src/gates/comparison.rs:325-402)"""
    g = GateAsm(pool)
    cb = -(-num_bits // num_chunks)
    nc = num_chunks
    for which in (0, 1):
        acc = g.reduce_with_powers([4 + which * nc + i for i in range(nc)], 1 << cb, wires=True)
        inp = g.wire(which)
        g.emit(g.sub(acc, inp, dst=acc))
        g.free(acc, inp)
    g.release()
    if cb > 4:
        raise ValueError("comparison chunks wider than 4 bits are not supported by this emitter")
    small = [g.imm(k) for k in range(1 << cb)]
    base = g.imm(1 << cb)
    one = small[1]
    msd = g.imm(0)
    for i in range(nc):
        f, s2 = g.wire(4 + i), g.wire(4 + nc + i)
        p = g.range_product(f, small)
        g.emit(p)
        g.free(p)
        p = g.range_product(s2, small)
        g.emit(p)
        g.free(p)
        diff = g.sub(s2, f)
        dummy, eq, inter = g.wire(4 + 2 * nc + i), g.wire(4 + 3 * nc + i), g.wire(4 + 4 * nc + i)
        neq = g.sub(one, eq)
        t = g.mul(diff, dummy)
        g.emit(g.sub(t, neq, dst=t))
        g.emit(g.mul(eq, diff, dst=t))
        g.mul(eq, msd, dst=t)
        g.emit(g.sub(inter, t, dst=t))
        g.mul(neq, diff, dst=t)
        g.add(inter, t, dst=msd)
        g.free(f, s2, diff, dummy, eq, inter, neq, t)
    w3 = g.wire(3)
    g.emit(g.sub(w3, msd, dst=msd))
    bits = [g.wire(4 + 5 * nc + i) for i in range(cb + 1)]
    for b in bits:
        t = g.sub(one, b)
        g.emit(g.mul(b, t, dst=t))
        g.free(t)
    comb = g.reduce_with_powers(bits, 2)
    t = g.add(w3, base)
    g.emit(g.sub(t, comb, dst=t))
    rb = g.wire(2)
    g.emit(g.sub(rb, bits[cb], dst=t))
    return g.instrs


def random_access_gate(bits, num_copies, num_extra_constants, pool):
    """RandomAccessGate (plonky2/src/gates/random_access.rs:409-450)"""
    g = GateAsm(pool)
    vs = 1 << bits
    routed = (2 + vs) * num_copies + num_extra_constants
    for c in range(num_copies):
        g.release()
        o = (2 + vs) * c
        one = g.imm(1)
        bs = [g.wire(routed + c * bits + i) for i in range(bits)]
        for b in bs:
            t = g.sub(b, one)
            g.emit(g.mul(b, t, dst=t))
            g.free(t)
        rec = g.imm(0)
        for b in reversed(bs):
            g.add(rec, rec, dst=rec)
            g.add(rec, b, dst=rec)
        idx = g.wire(o)
        g.emit(g.sub(rec, idx, dst=rec))
        g.free(rec, idx)
        items = [g.wire(o + 2 + i) for i in range(vs)]
        for b in bs:
            nxt = []
            for k in range(len(items) // 2):
                x, y = items[2 * k], items[2 * k + 1]
                g.sub(y, x, dst=y)
                g.mul(b, y, dst=y)
                g.add(x, y, dst=x)
                g.free(y)
                nxt.append(x)
            items = nxt
        claimed = g.wire(o + 1)
        g.emit(g.sub(items[0], claimed, dst=claimed))
    g.release()
    for i in range(num_extra_constants):
        c, w = g.const(i), g.wire((2 + vs) * num_copies + i)
        g.emit(g.sub(c, w, dst=c))
        g.free(c, w)
    return g.instrs


def poseidon_gate(pool):
    """PoseidonGate (plonky2/src/gates/poseidon.rs:485-564) over the permutation's own tables
    (hash/poseidon.rs:53-151, poseidon_goldilocks.rs:21-212)."""
    from .poseidon_tables import TABLES as T

    g = GateAsm(pool)
    SW, WIRE_SWAP, START_DELTA = 12, 24, 25
    START_FULL_0 = START_DELTA + 4
    START_PARTIAL = START_FULL_0 + SW * 3
    START_FULL_1 = START_PARTIAL + 22
    one = g.imm(1)
    swap = g.wire(WIRE_SWAP)
    t = g.sub(swap, one)
    g.emit(g.mul(swap, t, dst=t))
    g.free(t, one)
    state = [None] * SW
    for i in range(4):
        lhs, rhs, delta = g.wire(i), g.wire(i + 4), g.wire(START_DELTA + i)
        t = g.sub(rhs, lhs)
        g.mul(swap, t, dst=t)
        g.emit(g.sub(t, delta, dst=t))
        g.free(t)
        state[i] = g.add(lhs, delta, dst=lhs)
        state[i + 4] = g.sub(rhs, delta, dst=rhs)
        g.free(delta)
    g.free(swap)
    for i in range(8, SW):
        state[i] = g.wire(i)

    def constant_layer(rc):
        for i in range(SW):
            c = g.imm(T["ALL_ROUND_CONSTANTS"][rc * SW + i])
            g.add(state[i], c, dst=state[i])
            g.free(c)

    def sbox(x):
        x2 = g.mul(x, x)
        x4 = g.mul(x2, x2)
        g.mul(x, x2, dst=x2)
        g.mul(x2, x4, dst=x)
        g.free(x2, x4)

    def mds_layer():
        # row r = MDS_DIAG[r] * state[r] + sum_i MDS_CIRC[i] * state[(i + r) % 12]: thirteen weights below 64, so a
        # row is 26 multiply-adds and one fold instead of 13 modular multiplications and 12 modular additions
        new = []
        for r in range(SW):
            terms = [(state[(i + r) % SW], T["MDS_CIRC"][i]) for i in range(SW)]
            if T["MDS_DIAG"][r]:
                terms.append((state[r], T["MDS_DIAG"][r]))
            new.append(g.weighted_sum(terms))
        g.free(*state)
        state[:] = new

    def check_against_wire(i, wire):
        sin = g.wire(wire)
        g.emit(g.sub(state[i], sin, dst=state[i]))
        g.free(state[i])
        state[i] = sin

    rc = 0
    for r in range(4):
        constant_layer(rc)
        if r != 0:
            for i in range(SW):
                check_against_wire(i, START_FULL_0 + SW * (r - 1) + i)
        for i in range(SW):
            sbox(state[i])
        mds_layer()
        rc += 1
    # partial_first_constant_layer + mds_partial_layer_init
    for i in range(SW):
        c = g.imm(T["FAST_PARTIAL_FIRST_ROUND_CONSTANT"][i])
        g.add(state[i], c, dst=state[i])
        g.free(c)
    new = [state[0]]
    for c in range(1, SW):
        acc = g.imm(0)
        for r in range(1, SW):
            m = g.imm(T["FAST_PARTIAL_ROUND_INITIAL_MATRIX"][(r - 1) * 11 + (c - 1)])
            g.mul(state[r], m, dst=m)
            g.add(acc, m, dst=acc)
            g.free(m)
        new.append(acc)
    g.free(*state[1:])
    state[:] = new
    for r in range(22):
        check_against_wire(0, START_PARTIAL + r)
        sbox(state[0])
        if r < 21:
            c = g.imm(T["FAST_PARTIAL_ROUND_CONSTANTS"][r])
            g.add(state[0], c, dst=state[0])
            g.free(c)
        # mds_partial_layer_fast
        k = g.imm(T["MDS_CIRC"][0] + T["MDS_DIAG"][0])
        d = g.mul(state[0], k)
        g.free(k)
        for i in range(1, SW):
            wh = g.imm(T["FAST_PARTIAL_ROUND_W_HATS"][r * 11 + i - 1])
            g.mul(state[i], wh, dst=wh)
            g.add(d, wh, dst=d)
            g.free(wh)
        for i in range(1, SW):
            v = g.imm(T["FAST_PARTIAL_ROUND_VS"][r * 11 + i - 1])
            g.mul(state[0], v, dst=v)
            g.add(state[i], v, dst=state[i])
            g.free(v)
        g.free(state[0])
        state[0] = d
    rc += 22
    for r in range(4):
        constant_layer(rc)
        for i in range(SW):
            check_against_wire(i, START_FULL_1 + SW * r + i)
        for i in range(SW):
            sbox(state[i])
        mds_layer()
        rc += 1
    for i in range(SW):
        out = g.wire(SW + i)
        g.emit(g.sub(state[i], out, dst=out))
        g.free(out)
    return g.instrs


# ---- the gates of upstream plonky2 beyond the ed25519 list (round 5) -------------------------------------------------------------
# Extension-field gates see pairs of wires as elements of F_p[X]/(X^2 - 7) (EvaluationVarsBase::get_local_ext, plonk/vars.rs:122-129;
# field/src/extension/quadratic.rs:173-185). A pair is a Python tuple (register of c0, register of c1).
EXT_W = 7


def _ext_wire(g, at):
    return (g.wire(at), g.wire(at + 1))


def _ext_mul(g, x, y):
    """(x0 + x1 X)(y0 + y1 X): c0 = x0 y0 + 7 x1 y1 through an accumulator (weights 1 and 7, one fold), c1 = x0 y1 + x1 y0"""
    t00, t11 = g.mul(x[0], y[0]), g.mul(x[1], y[1])
    c0 = g.weighted_sum([(t00, 1), (t11, EXT_W)])
    g.free(t00, t11)
    t01, t10 = g.mul(x[0], y[1]), g.mul(x[1], y[0])
    c1 = g.add(t01, t10, dst=t01)
    g.free(t10)
    return (c0, c1)


def _ext_free(g, *pairs):
    for x in pairs:
        g.free(x[0], x[1])


def _ext_emit_diff(g, a, b):
    """emit a - b component by component (to_basefield_array); `a` is overwritten"""
    g.emit(g.sub(a[0], b[0], dst=a[0]))
    g.emit(g.sub(a[1], b[1], dst=a[1]))


def arithmetic_extension_gate(num_ops, pool):
    """ArithmeticExtensionGate { num_ops } (plonky2/src/gates/arithmetic_extension.rs:129-147)"""
    g = GateAsm(pool)
    for i in range(num_ops):
        g.release()
        c0, c1 = g.const(0), g.const(1)
        m0, m1, ad, out = (_ext_wire(g, 8 * i + 2 * k) for k in range(4))
        m = _ext_mul(g, m0, m1)
        for k in (0, 1):  # computed = m * c0 + addend * c1
            g.mul(m[k], c0, dst=m[k])
            g.mul(ad[k], c1, dst=ad[k])
            g.add(m[k], ad[k], dst=m[k])
        _ext_emit_diff(g, out, m)
    return g.instrs


def mul_extension_gate(num_ops, pool):
    """MulExtensionGate { num_ops } (plonky2/src/gates/multiplication_extension.rs:122-137)"""
    g = GateAsm(pool)
    for i in range(num_ops):
        g.release()
        c0 = g.const(0)
        m0, m1, out = (_ext_wire(g, 6 * i + 2 * k) for k in range(3))
        m = _ext_mul(g, m0, m1)
        for k in (0, 1):
            g.mul(m[k], c0, dst=m[k])
        _ext_emit_diff(g, out, m)
    return g.instrs


def reducing_gate(num_coeffs, pool, extension_coeffs=False):
    """ReducingGate { num_coeffs } (plonky2/src/gates/reducing.rs:160-181) and, with extension_coeffs, ReducingExtensionGate
    (reducing_extension.rs:157-178): acc_i = acc_{i-1} * alpha + coeff_i, the last accumulator being the output wires."""
    g = GateAsm(pool)
    d = 2
    start_coeffs = 3 * d
    start_accs = start_coeffs + (d * num_coeffs if extension_coeffs else num_coeffs)
    alpha = _ext_wire(g, d)
    acc = _ext_wire(g, 2 * d)
    for i in range(num_coeffs):
        t = _ext_mul(g, acc, alpha)
        _ext_free(g, acc)
        if extension_coeffs:
            c = _ext_wire(g, start_coeffs + d * i)
            g.add(t[0], c[0], dst=t[0])
            g.add(t[1], c[1], dst=t[1])
            _ext_free(g, c)
        else:
            c = g.wire(start_coeffs + i)
            g.add(t[0], c, dst=t[0])
            g.free(c)
        nxt = _ext_wire(g, 0 if i == num_coeffs - 1 else start_accs + d * i)
        _ext_emit_diff(g, t, nxt)
        _ext_free(g, t)
        acc = nxt
    return g.instrs


def exponentiation_gate(num_power_bits, pool):
    """ExponentiationGate { num_power_bits } (plonky2/src/gates/exponentiation.rs:266-298)"""
    g = GateAsm(pool)
    n = num_power_bits
    one = g.imm(1)
    base = g.wire(0)
    prev = None
    for i in range(n):
        cur_bit = g.wire(1 + (n - 1 - i))  # power bits are little-endian, accumulated big-endian
        t = g.mul(cur_bit, base)
        g.add(t, one, dst=t)
        g.sub(t, cur_bit, dst=t)  # cur_bit * base + (1 - cur_bit)
        g.free(cur_bit)
        if prev is not None:
            g.mul(prev, prev, dst=prev)
            g.mul(prev, t, dst=t)
            g.free(prev)
        inter = g.wire(2 + n + i)
        g.emit(g.sub(t, inter, dst=t))
        g.free(t)
        prev = inter
    out = g.wire(1 + n)
    g.emit(g.sub(out, prev, dst=out))
    return g.instrs


def poseidon_mds_gate(pool):
    """PoseidonMdsGate (plonky2/src/gates/poseidon_mds.rs:184-204): outputs = mds_layer_field(inputs) over F_p^2 — every component a sum
    of thirteen small multiples, i.e. one accumulator fold each"""
    from .poseidon_tables import TABLES as T

    g = GateAsm(pool)
    SW = 12
    ins = [_ext_wire(g, 2 * i) for i in range(SW)]
    for r in range(SW):
        for k in (0, 1):
            terms = [(ins[(i + r) % SW][k], T["MDS_CIRC"][i]) for i in range(SW)]
            if T["MDS_DIAG"][r]:
                terms.append((ins[r][k], T["MDS_DIAG"][r]))
            computed = g.weighted_sum(terms)
            out = g.wire(2 * (SW + r) + k)
            g.emit(g.sub(out, computed, dst=out))
            g.free(out, computed)
    return g.instrs


def _root_of_unity(bits):
    return pow(1753635133440165772, 1 << (32 - bits), P)  # F::primitive_root_of_unity (field/src/types.rs:268-272)


def interpolation_gate(subgroup_bits, pool, low_degree):
    """HighDegreeInterpolationGate (plonky2/src/gates/high_degree_interpolation.rs:119-147) / LowDegreeInterpolationGate
    (low_degree_interpolation.rs:356-404); wire layout gates/interpolation.rs:19-76."""
    g = GateAsm(pool)
    d, np_ = 2, 1 << subgroup_bits
    if np_ > 16:
        raise ValueError("interpolation gates with more than 16 points are not supported by this emitter")
    start_values, eval_point, eval_value = 1, 1 + np_ * d, 1 + np_ * d + d
    start_coeffs = eval_value + d
    end_coeffs = start_coeffs + np_ * d
    w = _root_of_unity(subgroup_bits)
    shift = g.wire(0)

    def eval_base(cs, x):
        """Horner with a base-field point in register x: acc = acc * x + c, component-wise; returns a fresh pair"""
        z = g.imm(0)
        acc = (g.add(cs[-1][0], z), g.add(cs[-1][1], z))  # copies of the leading coefficient (ADD with a zero immediate)
        g.free(z)
        for c in reversed(cs[:-1]):
            for k in (0, 1):
                g.mul(acc[k], x, dst=acc[k])
                g.add(acc[k], c[k], dst=acc[k])
        return acc

    coeffs = [_ext_wire(g, start_coeffs + d * i) for i in range(np_)]
    if low_degree:
        # powers of the shift: wire i (i = 2..np-1) must be shift^(i-1) * shift; coefficient i is altered by shift^i on the way, so that
        # at most two powers are live at a time (altered_coeffs[i] = c_i * shift^i, then altered(w^i) = original(shift * w^i))
        prev = shift
        for i in range(1, np_):
            for k in (0, 1):
                g.mul(coeffs[i][k], prev, dst=coeffs[i][k])
            if i < np_ - 1:
                nxt = g.wire(end_coeffs + i - 1)
                t = g.mul(prev, shift)
                g.emit(g.sub(t, nxt, dst=t))
                g.free(t)
                if prev != shift:
                    g.free(prev)
                prev = nxt
        if prev != shift:
            g.free(prev)
    for i in range(np_):
        x = g.imm(pow(w, i, P))
        if not low_degree:
            g.mul(x, shift, dst=x)  # coset(shift) = g^i * shift
        computed = eval_base(coeffs, x)
        g.free(x)
        value = _ext_wire(g, start_values + d * i)
        _ext_emit_diff(g, value, computed)
        _ext_free(g, value, computed)
    ep = _ext_wire(g, eval_point)
    if low_degree:
        _ext_free(g, *coeffs)
        prev = ep
        for i in range(1, np_ - 1):  # powers of the evaluation point: wire pair i+1 must be (pair i) * point
            nxt = _ext_wire(g, end_coeffs + np_ - 2 + (i - 1) * d)
            t = _ext_mul(g, prev, ep)
            _ext_emit_diff(g, t, nxt)
            _ext_free(g, t)
            if prev is not ep:
                _ext_free(g, prev)
            prev = nxt
        if prev is not ep:
            _ext_free(g, prev)
        acc = _ext_wire(g, start_coeffs)  # eval_with_powers uses the ORIGINAL coefficients: c_0 + sum c_i * point^i
        for i in range(1, np_):
            pw = ep if i == 1 else _ext_wire(g, end_coeffs + np_ - 2 + (i - 2) * d)
            c = _ext_wire(g, start_coeffs + d * i)
            t = _ext_mul(g, pw, c)
            g.add(acc[0], t[0], dst=acc[0])
            g.add(acc[1], t[1], dst=acc[1])
            _ext_free(g, t, c)
            if pw is not ep:
                _ext_free(g, pw)
    else:
        acc = coeffs[-1]  # interpolant.eval(point): Horner over the extension
        for c in reversed(coeffs[:-1]):
            t = _ext_mul(g, acc, ep)
            g.add(t[0], c[0], dst=t[0])
            g.add(t[1], c[1], dst=t[1])
            _ext_free(g, acc)
            acc = t
    value = _ext_wire(g, eval_value)
    _ext_emit_diff(g, value, acc)
    return g.instrs


def build_gate(kind, param, pool):
    """(kind, param) -> instruction list; the kinds of the ed25519 gate list and, since round 5, the other gates of
    upstream plonky2 that standard_recursion_config circuits are made of"""
    if kind == "noop":
        return noop_gate()
    if kind == "constant":
        return constant_gate(param)
    if kind == "public_input":
        return public_input_gate()
    if kind == "arithmetic":
        return arithmetic_gate(param)
    if kind == "base_sum":
        return base_sum_gate(param[0], param[1], pool)
    if kind == "u32_add_many":
        return u32_add_many_gate(param[0], param[1], pool)
    if kind == "u32_arithmetic":
        return u32_arithmetic_gate(param, pool)
    if kind == "u32_subtraction":
        return u32_subtraction_gate(param, pool)
    if kind == "u32_range_check":
        return u32_range_check_gate(param, pool)
    if kind == "comparison":
        return comparison_gate(param[0], param[1], pool)
    if kind == "random_access":
        return random_access_gate(param[0], param[1], param[2], pool)
    if kind == "poseidon":
        return poseidon_gate(pool)
    if kind == "arithmetic_extension":
        return arithmetic_extension_gate(param, pool)
    if kind == "mul_extension":
        return mul_extension_gate(param, pool)
    if kind == "reducing":
        return reducing_gate(param, pool)
    if kind == "reducing_extension":
        return reducing_gate(param, pool, extension_coeffs=True)
    if kind == "exponentiation":
        return exponentiation_gate(param, pool)
    if kind == "poseidon_mds":
        return poseidon_mds_gate(pool)
    if kind == "low_degree_interpolation":
        return interpolation_gate(param, pool, low_degree=True)
    if kind == "high_degree_interpolation":
        return interpolation_gate(param, pool, low_degree=False)
    raise ValueError(f"no register-program emitter for gate kind {kind!r}")


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
