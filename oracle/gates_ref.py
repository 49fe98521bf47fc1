"""gates_ref.py — CPU restatement of the gate constraints of the ed25519 circuit's gate list
 beyond the four in plonk_ref.py (TEST INFRASTRUCTURE ONLY), with witness
generators for single rows. Each constraint function is written once over an abstract field `F`
(base field or F_{p^2}) the way the reference writes `eval_unfiltered` / `eval_unfiltered_base_*`
from one definition.

Parity status: the reference's own gate tests check (a) eval_unfiltered == eval_unfiltered_base
and low degree (gates/gate_testing.rs) and (b) that an honestly generated row satisfies the
constraints (e.g. u32/src/gates/arithmetic_u32.rs tests, gates/poseidon.rs `generated_output`).
tests/test_oracle_gates.py does both here: rows produced by the generators below (restated from
each gate's `run_once`) satisfy every constraint over the base field AND over the extension,
single-wire corruptions violate them, and the Poseidon gate's output wires equal the permutation
pinned by the reference's known answers.
"""
from . import pyref

P = pyref.P
W = 7


class Base:
    zero, one = 0, 1

    @staticmethod
    def c(x):
        return x % P

    @staticmethod
    def add(a, b):
        return (a + b) % P

    @staticmethod
    def sub(a, b):
        return (a - b) % P

    @staticmethod
    def mul(a, b):
        return a * b % P


class Ext:
    zero, one = (0, 0), (1, 0)

    @staticmethod
    def c(x):
        return (x % P, 0)

    @staticmethod
    def add(a, b):
        return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)

    @staticmethod
    def sub(a, b):
        return ((a[0] - b[0]) % P, (a[1] - b[1]) % P)

    @staticmethod
    def mul(a, b):
        return ((a[0] * b[0] + W * a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)


def _reduce_with_powers(F, terms, base):
    """plonk_common.rs:116-128: sum terms[i] * base^i"""
    acc = F.zero
    for t in reversed(terms):
        acc = F.add(F.mul(acc, F.c(base)), t)
    return acc


def _range_product(F, x, bound):
    """prod_{k < bound} (x - k)"""
    acc = F.one
    for k in range(bound):
        acc = F.mul(acc, F.sub(x, F.c(k)))
    return acc


# ------------------------------------------------------------------------------------------------
def num_wires(kind, param):
    if kind == "base_sum":
        return 1 + param[1]
    if kind == "u32_add_many":
        na, ops = param
        return (na + 3) * ops + 18 * ops
    if kind == "u32_arithmetic":
        return 6 * param + 32 * param
    if kind == "u32_subtraction":
        return 5 * param + 16 * param
    if kind == "u32_range_check":
        return param + 16 * param
    if kind == "comparison":
        nb, nc = param
        return 4 + 5 * nc + (-(-nb // nc) + 1)
    if kind == "random_access":
        bits, copies, extra = param
        return (2 + (1 << bits)) * copies + extra + copies * bits
    if kind == "poseidon":
        return 25 + 4 + 12 * 3 + 22 + 12 * 4
    return {"noop": 0, "constant": param, "public_input": 4, "arithmetic": 4 * (param or 0)}[kind]


def num_constraints(kind, param):
    if kind == "base_sum":
        return 1 + param[1]
    if kind == "u32_add_many":
        return param[1] * (3 + 18)
    if kind == "u32_arithmetic":
        return param * (4 + 32)
    if kind == "u32_subtraction":
        return param * (3 + 16)
    if kind == "u32_range_check":
        return param * 17
    if kind == "comparison":
        nb, nc = param
        return 6 + 5 * nc + -(-nb // nc)
    if kind == "random_access":
        bits, copies, extra = param
        return copies * (bits + 2) + extra
    if kind == "poseidon":
        return 12 * 7 + 22 + 12 + 1 + 4
    return {"noop": 0, "constant": param, "public_input": 4, "arithmetic": param}[kind]


def constraints(kind, param, consts, w, pih, F):
    """Gate::eval_unfiltered / eval_unfiltered_base_* for one point: consts = local_constants after the
    selector prefix, w = local_wires, pih = public_inputs_hash (base elements)."""
    out = []
    if kind == "noop":
        return out
    if kind == "constant":  # gates/constant.rs:150-158
        return [F.sub(consts[i], w[i]) for i in range(param)]
    if kind == "public_input":  # gates/public_input.rs:129-139
        return [F.sub(w[i], F.c(pih[i])) for i in range(4)]
    if kind == "arithmetic":  # gates/arithmetic_base.rs:199-216
        for i in range(param):
            computed = F.add(F.mul(F.mul(w[4 * i], w[4 * i + 1]), consts[0]), F.mul(w[4 * i + 2], consts[1]))
            out.append(F.sub(w[4 * i + 3], computed))
        return out
    if kind == "base_sum":  # gates/base_sum.rs:213-230
        B, nl = param
        limbs = w[1 : 1 + nl]
        out.append(F.sub(_reduce_with_powers(F, limbs, B), w[0]))
        out += [_range_product(F, limb, B) for limb in limbs]
        return out
    if kind == "u32_add_many":  # u32/src/gates/add_many_u32.rs:143-184
        na, ops = param
        for i in range(ops):
            o = (na + 3) * i
            computed = w[o + na]  # carry
            for j in range(na):
                computed = F.add(computed, w[o + j])
            res, car = w[o + na + 1], w[o + na + 2]
            out.append(F.sub(F.add(F.mul(car, F.c(1 << 32)), res), computed))
            comb_res, comb_car = F.zero, F.zero
            for j in reversed(range(18)):
                limb = w[(na + 3) * ops + 18 * i + j]
                out.append(_range_product(F, limb, 4))
                if j < 16:
                    comb_res = F.add(F.mul(F.c(4), comb_res), limb)
                else:
                    comb_car = F.add(F.mul(F.c(4), comb_car), limb)
            out.append(F.sub(comb_res, res))
            out.append(F.sub(comb_car, car))
        return out
    if kind == "u32_arithmetic":  # u32/src/gates/arithmetic_u32.rs:326-385
        ops = param
        for i in range(ops):
            m0, m1, ad, lo, hi, inv = (w[6 * i + k] for k in range(6))
            computed = F.add(F.mul(m0, m1), ad)
            diff = F.sub(F.c(0xFFFFFFFF), hi)
            hi_not_max = F.sub(F.mul(inv, diff), F.one)
            out.append(F.mul(hi_not_max, lo))
            out.append(F.sub(F.add(F.mul(hi, F.c(1 << 32)), lo), computed))
            c_lo, c_hi = F.zero, F.zero
            for j in reversed(range(32)):
                limb = w[6 * ops + 32 * i + j]
                out.append(_range_product(F, limb, 4))
                if j < 16:
                    c_lo = F.add(F.mul(c_lo, F.c(4)), limb)
                else:
                    c_hi = F.add(F.mul(c_hi, F.c(4)), limb)
            out.append(F.sub(c_lo, lo))
            out.append(F.sub(c_hi, hi))
        return out
    if kind == "u32_subtraction":  # u32/src/gates/subtraction_u32.rs:233-269
        ops = param
        for i in range(ops):
            x, y, bi, res, bo = (w[5 * i + k] for k in range(5))
            initial = F.sub(F.sub(x, y), bi)
            out.append(F.sub(res, F.add(initial, F.mul(bo, F.c(1 << 32)))))
            comb = F.zero
            for j in reversed(range(16)):
                limb = w[5 * ops + 16 * i + j]
                out.append(_range_product(F, limb, 4))
                comb = F.add(F.mul(comb, F.c(4)), limb)
            out.append(F.sub(comb, res))
            out.append(F.mul(bo, F.sub(F.one, bo)))
        return out
    if kind == "u32_range_check":  # u32/src/gates/range_check_u32.rs:89-111
        nl = param
        for i in range(nl):
            aux = [w[nl + 16 * i + j] for j in range(16)]
            out.append(F.sub(_reduce_with_powers(F, aux, 4), w[i]))
            out += [_range_product(F, a, 4) for a in aux]
        return out
    if kind == "comparison":  # u32/src/gates/comparison.rs:325-402
        nb, nc = param
        cb = -(-nb // nc)
        first = [w[4 + i] for i in range(nc)]
        second = [w[4 + nc + i] for i in range(nc)]
        out.append(F.sub(_reduce_with_powers(F, first, 1 << cb), w[0]))
        out.append(F.sub(_reduce_with_powers(F, second, 1 << cb), w[1]))
        msd = F.zero
        for i in range(nc):
            out.append(_range_product(F, first[i], 1 << cb))
            out.append(_range_product(F, second[i], 1 << cb))
            diff = F.sub(second[i], first[i])
            dummy, eq, inter = w[4 + 2 * nc + i], w[4 + 3 * nc + i], w[4 + 4 * nc + i]
            out.append(F.sub(F.mul(diff, dummy), F.sub(F.one, eq)))
            out.append(F.mul(eq, diff))
            out.append(F.sub(inter, F.mul(eq, msd)))
            msd = F.add(inter, F.mul(F.sub(F.one, eq), diff))
        out.append(F.sub(w[3], msd))
        bits = [w[4 + 5 * nc + i] for i in range(cb + 1)]
        out += [F.mul(b, F.sub(F.one, b)) for b in bits]
        out.append(F.sub(F.add(w[3], F.c(1 << cb)), _reduce_with_powers(F, bits, 2)))
        out.append(F.sub(w[2], bits[cb]))
        return out
    if kind == "random_access":  # gates/random_access.rs:409-450
        bits_n, copies, extra = param
        vs = 1 << bits_n
        routed = (2 + vs) * copies + extra
        for c in range(copies):
            o = (2 + vs) * c
            idx, claimed = w[o], w[o + 1]
            items = [w[o + 2 + i] for i in range(vs)]
            bits = [w[routed + c * bits_n + i] for i in range(bits_n)]
            out += [F.mul(b, F.sub(b, F.one)) for b in bits]
            rec = F.zero
            for b in reversed(bits):
                rec = F.add(F.add(rec, rec), b)
            out.append(F.sub(rec, idx))
            for b in bits:
                items = [F.add(items[2 * k], F.mul(b, F.sub(items[2 * k + 1], items[2 * k]))) for k in range(len(items) // 2)]
            out.append(F.sub(items[0], claimed))
        out += [F.sub(consts[i], w[(2 + vs) * copies + i]) for i in range(extra)]
        return out
    if kind == "poseidon":  # gates/poseidon.rs:485-564
        return _poseidon_gate(F, w)
    raise ValueError(kind)


# ---- Poseidon gate -----------------------------------------------------------------------------
_C = pyref._C
SW = 12
WIRE_SWAP, START_DELTA = 24, 25
START_FULL_0 = START_DELTA + 4
START_PARTIAL = START_FULL_0 + SW * 3
START_FULL_1 = START_PARTIAL + 22


def _pos_constant_layer(F, s, rc):
    return [F.add(x, F.c(_C["POSEIDON_ALL_ROUND_CONSTANTS"][rc * SW + i])) for i, x in enumerate(s)]


def _pos_sbox(F, x):
    x2 = F.mul(x, x)
    x4 = F.mul(x2, x2)
    return F.mul(F.mul(x, x2), x4)


def _pos_mds(F, s):
    out = []
    for r in range(SW):
        acc = F.mul(s[r], F.c(_C["POSEIDON_MDS_DIAG"][r]))
        for i in range(SW):
            acc = F.add(acc, F.mul(s[(i + r) % SW], F.c(_C["POSEIDON_MDS_CIRC"][i])))
        out.append(acc)
    return out


def _pos_partial_init(F, s):
    """partial_first_constant_layer + mds_partial_layer_init (hash/poseidon.rs:312-365)"""
    s = [F.add(x, F.c(_C["POSEIDON_FAST_PARTIAL_FIRST_ROUND_CONSTANT"][i])) for i, x in enumerate(s)]
    M = _C["POSEIDON_FAST_PARTIAL_ROUND_INITIAL_MATRIX"]
    out = [s[0]]
    for c in range(1, SW):
        acc = F.zero
        for r in range(1, SW):
            acc = F.add(acc, F.mul(s[r], F.c(M[(r - 1) * 11 + (c - 1)])))
        out.append(acc)
    return out


def _pos_partial_fast(F, s, r):
    """mds_partial_layer_fast (hash/poseidon.rs:400-427)"""
    wh, vs = _C["POSEIDON_FAST_PARTIAL_ROUND_W_HATS"], _C["POSEIDON_FAST_PARTIAL_ROUND_VS"]
    d = F.mul(s[0], F.c(_C["POSEIDON_MDS_CIRC"][0] + _C["POSEIDON_MDS_DIAG"][0]))
    for i in range(1, SW):
        d = F.add(d, F.mul(s[i], F.c(wh[r * 11 + i - 1])))
    return [d] + [F.add(s[i], F.mul(s[0], F.c(vs[r * 11 + i - 1]))) for i in range(1, SW)]


def _poseidon_gate(F, w):
    out = []
    swap = w[WIRE_SWAP]
    out.append(F.mul(swap, F.sub(swap, F.one)))
    for i in range(4):
        out.append(F.sub(F.mul(swap, F.sub(w[i + 4], w[i])), w[START_DELTA + i]))
    s = [None] * SW
    for i in range(4):
        s[i] = F.add(w[i], w[START_DELTA + i])
        s[i + 4] = F.sub(w[i + 4], w[START_DELTA + i])
    for i in range(8, SW):
        s[i] = w[i]
    rc = 0
    for r in range(4):
        s = _pos_constant_layer(F, s, rc)
        if r != 0:
            for i in range(SW):
                sin = w[START_FULL_0 + SW * (r - 1) + i]
                out.append(F.sub(s[i], sin))
                s[i] = sin
        s = _pos_mds(F, [_pos_sbox(F, x) for x in s])
        rc += 1
    s = _pos_partial_init(F, s)
    for r in range(22):
        sin = w[START_PARTIAL + r]
        out.append(F.sub(s[0], sin))
        s[0] = _pos_sbox(F, sin)
        if r < 21:
            s[0] = F.add(s[0], F.c(_C["POSEIDON_FAST_PARTIAL_ROUND_CONSTANTS"][r]))
        s = _pos_partial_fast(F, s, r)
    rc += 22
    for r in range(4):
        s = _pos_constant_layer(F, s, rc)
        for i in range(SW):
            sin = w[START_FULL_1 + SW * r + i]
            out.append(F.sub(s[i], sin))
            s[i] = sin
        s = _pos_mds(F, [_pos_sbox(F, x) for x in s])
        rc += 1
    for i in range(SW):
        out.append(F.sub(s[i], w[SW + i]))
    return out


# ---- witness rows (each gate's generator, restated) --------------------------------------------
def _digits(x, base, n):
    out = []
    for _ in range(n):
        out.append(x % base)
        x //= base
    return out


def fill_row(kind, param, rng, consts, pih):
    """An honestly generated row: list of num_wires(kind, param) values satisfying constraints()."""
    F = Base
    if kind == "noop":
        return []
    if kind == "constant":
        return [consts[i] for i in range(param)]
    if kind == "public_input":
        return list(pih)
    if kind == "arithmetic":
        w = []
        for _ in range(param):
            m0, m1, ad = (rng.randrange(P) for _ in range(3))
            w += [m0, m1, ad, (m0 * m1 % P * consts[0] + ad * consts[1]) % P]
        return w
    if kind == "base_sum":  # BaseSplitGenerator (gates/base_sum.rs:233-270)
        B, nl = param
        x = rng.randrange(B ** nl)
        return [x] + _digits(x, B, nl)
    if kind == "u32_add_many":  # U32AddManyGenerator (add_many_u32.rs:280-340)
        na, ops = param
        routed, limbs = [], []
        for _ in range(ops):
            addends = [rng.randrange(1 << 32) for _ in range(na)]
            carry = rng.randrange(1 << 32)
            total = sum(addends) + carry
            res, car = total & 0xFFFFFFFF, total >> 32
            routed += addends + [carry, res, car]
            limbs += _digits(res, 4, 16) + _digits(car, 4, 2)
        return routed + limbs
    if kind == "u32_arithmetic":  # U32ArithmeticGenerator (arithmetic_u32.rs:410-460)
        routed, limbs = [], []
        for _ in range(param):
            m0, m1, ad = (rng.randrange(1 << 32) for _ in range(3))
            o = m0 * m1 + ad
            hi, lo = o >> 32, o & 0xFFFFFFFF
            diff = 0xFFFFFFFF - hi
            routed += [m0, m1, ad, lo, hi, pow(diff, P - 2, P) if diff else 0]
            limbs += _digits(o, 4, 32)
        return routed + limbs
    if kind == "u32_subtraction":  # U32SubtractionGenerator (subtraction_u32.rs:293-338)
        routed, limbs = [], []
        for _ in range(param):
            x, y, bi = rng.randrange(1 << 32), rng.randrange(1 << 32), rng.randrange(2)
            initial = (x - y - bi) % P
            bo = 1 if initial > (1 << 32) else 0
            res = (initial + (bo << 32)) % P
            routed += [x, y, bi, res, bo]
            limbs += _digits(res, 4, 16)
        return routed + limbs
    if kind == "u32_range_check":  # U32RangeCheckGenerator (range_check_u32.rs:188-210)
        vals = [rng.randrange(1 << 32) for _ in range(param)]
        return vals + [d for v in vals for d in _digits(v, 4, 16)]
    if kind == "comparison":  # ComparisonGenerator (comparison.rs:423-521)
        nb, nc = param
        cb = -(-nb // nc)
        a, b = rng.randrange(1 << nb), rng.randrange(1 << nb)
        if rng.randrange(4) == 0:
            b = a
        fc, sc = _digits(a, 1 << cb, nc), _digits(b, 1 << cb, nc)
        eq = [int(x == y) for x, y in zip(fc, sc)]
        dummy = [1 if x == y else pow((y - x) % P, P - 2, P) for x, y in zip(fc, sc)]
        msd, inter = 0, []
        for x, y in zip(fc, sc):
            if x != y:
                msd = (y - x) % P
                inter.append(0)
            else:
                inter.append(msd)
        bits = _digits(((1 << cb) + msd) % P, 2, cb + 1)
        return [a, b, int(a <= b), msd] + fc + sc + dummy + eq + inter + bits
    if kind == "random_access":  # RandomAccessGenerator (random_access.rs:473-510)
        bits_n, copies, extra = param
        vs = 1 << bits_n
        routed, bit_wires = [], []
        for _ in range(copies):
            idx = rng.randrange(vs)
            items = [rng.randrange(P) for _ in range(vs)]
            routed += [idx, items[idx]] + items
            bit_wires += _digits(idx, 2, bits_n)
        return routed + [consts[i] for i in range(extra)] + bit_wires
    if kind == "poseidon":  # PoseidonGenerator (gates/poseidon.rs:760-860)
        inputs = [rng.randrange(P) for _ in range(SW)]
        swap = rng.randrange(2)
        w = [0] * num_wires(kind, param)
        w[:SW] = inputs
        w[WIRE_SWAP] = swap
        s = list(inputs)
        for i in range(4):
            delta = swap * (inputs[i + 4] - inputs[i]) % P
            w[START_DELTA + i] = delta
            s[i], s[i + 4] = (inputs[i] + delta) % P, (inputs[i + 4] - delta) % P
        expected = pyref.poseidon(s)
        rc = 0
        for r in range(4):
            s = _pos_constant_layer(F, s, rc)
            if r != 0:
                for i in range(SW):
                    w[START_FULL_0 + SW * (r - 1) + i] = s[i]
            s = _pos_mds(F, [_pos_sbox(F, x) for x in s])
            rc += 1
        s = _pos_partial_init(F, s)
        for r in range(22):
            w[START_PARTIAL + r] = s[0]
            s[0] = _pos_sbox(F, s[0])
            if r < 21:
                s[0] = (s[0] + _C["POSEIDON_FAST_PARTIAL_ROUND_CONSTANTS"][r]) % P
            s = _pos_partial_fast(F, s, r)
        rc += 22
        for r in range(4):
            s = _pos_constant_layer(F, s, rc)
            for i in range(SW):
                w[START_FULL_1 + SW * r + i] = s[i]
            s = _pos_mds(F, [_pos_sbox(F, x) for x in s])
            rc += 1
        w[SW : 2 * SW] = s
        assert s == expected, "the gate's fast partial rounds must equal the textbook permutation"
        return w
    raise ValueError(kind)
