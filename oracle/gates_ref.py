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
    if kind == "arithmetic_extension":
        return 4 * D * param
    if kind == "mul_extension":
        return 3 * D * param
    if kind == "reducing":
        return 3 * D + param + D * (param - 1)
    if kind == "reducing_extension":
        return 3 * D + D * param + D * (param - 1)
    if kind == "exponentiation":
        return 2 + 2 * param
    if kind == "poseidon_mds":
        return 2 * SW * D
    if kind == "high_degree_interpolation":
        return 1 + (1 << param) * D + 2 * D + (1 << param) * D
    if kind == "low_degree_interpolation":
        np_ = 1 << param
        return 1 + np_ * D + 2 * D + np_ * D + (np_ - 2) + (np_ - 2) * D
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
    if kind in ("arithmetic_extension", "mul_extension", "reducing", "reducing_extension"):
        return D * param  # arithmetic_extension.rs:205, multiplication_extension.rs:191, reducing.rs:235, reducing_extension.rs:233
    if kind == "exponentiation":
        return param + 1  # exponentiation.rs:258
    if kind == "poseidon_mds":
        return SW * D  # poseidon_mds.rs:246
    if kind == "high_degree_interpolation":
        return (1 << param) * D + D  # high_degree_interpolation.rs:207-211
    if kind == "low_degree_interpolation":
        return (1 << param) * D + D + (D + 1) * ((1 << param) - 2)  # low_degree_interpolation.rs:505-510
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
    A = Algebra(F)
    if kind == "arithmetic_extension":  # gates/arithmetic_extension.rs:110-147; wires 4*D*i + {0, D, 2D, 3D}
        for i in range(param):
            m0, m1, ad, o = (A.get(w, 4 * D * i + D * k) for k in range(4))
            computed = A.add(A.scalar(A.mul(m0, m1), consts[0]), A.scalar(ad, consts[1]))
            out += A.sub(o, computed)
        return out
    if kind == "mul_extension":  # gates/multiplication_extension.rs:103-137; wires 3*D*i + {0, D, 2D}
        for i in range(param):
            m0, m1, o = (A.get(w, 3 * D * i + D * k) for k in range(3))
            out += A.sub(o, A.scalar(A.mul(m0, m1), consts[0]))
        return out
    if kind in ("reducing", "reducing_extension"):  # gates/reducing.rs:160-181, reducing_extension.rs:157-178
        nc = param
        ext_coeffs = kind == "reducing_extension"
        alpha, acc = A.get(w, D), A.get(w, 2 * D)
        start_coeffs = 3 * D
        start_accs = start_coeffs + (D * nc if ext_coeffs else nc)
        for i in range(nc):
            nxt = A.get(w, 0) if i == nc - 1 else A.get(w, start_accs + D * i)  # the last accumulator is the output
            coeff = A.get(w, start_coeffs + D * i) if ext_coeffs else A.from_base(w[start_coeffs + i])
            out += A.sub(A.add(A.mul(acc, alpha), coeff), nxt)
            acc = nxt
        return out
    if kind == "exponentiation":  # gates/exponentiation.rs:266-298
        n = param
        base, output = w[0], w[1 + n]
        bits, inter = [w[1 + i] for i in range(n)], [w[2 + n + i] for i in range(n)]
        for i in range(n):
            prev = F.one if i == 0 else F.mul(inter[i - 1], inter[i - 1])
            cur_bit = bits[n - 1 - i]  # power_bits is in LE order, but we accumulate in BE order
            computed = F.mul(prev, F.add(F.mul(cur_bit, base), F.sub(F.one, cur_bit)))
            out.append(F.sub(computed, inter[i]))
        out.append(F.sub(output, inter[n - 1]))
        return out
    if kind == "poseidon_mds":  # gates/poseidon_mds.rs:184-204 with mds_layer_field / mds_layer_algebra (:49-110)
        inputs = [A.get(w, D * i) for i in range(SW)]
        for r in range(SW):
            acc = A.scalar(inputs[r], F.c(_C["POSEIDON_MDS_DIAG"][r]))
            for i in range(SW):
                acc = A.add(acc, A.scalar(inputs[(i + r) % SW], F.c(_C["POSEIDON_MDS_CIRC"][i])))
            out += A.sub(A.get(w, D * (SW + r)), acc)
        return out
    if kind in ("high_degree_interpolation", "low_degree_interpolation"):
        return _interpolation_gate(F, A, kind == "low_degree_interpolation", param, w)
    raise ValueError(kind)


D = 2  # the extension degree the gates are instantiated with (GoldilocksField: Extendable<2>)


class Algebra:
    """D = 2 "extension algebra" over F (field/src/extension/algebra.rs): pairs [a0, a1] of F elements meaning a0 + a1 X with
    X^2 = W. With F = Base this is F_p^2 itself (EvaluationVarsBase::get_local_ext, plonk/vars.rs:122-129); with F = Ext it is
    ExtensionAlgebra<F_p^2, 2>, what eval_unfiltered uses for the same wires (get_local_ext_algebra)."""

    def __init__(self, F):
        self.F = F
        self.W = F.c(W)

    def get(self, w, at):
        return [w[at], w[at + 1]]

    def from_base(self, x):
        return [x, self.F.zero]

    def add(self, x, y):
        return [self.F.add(x[0], y[0]), self.F.add(x[1], y[1])]

    def sub(self, x, y):
        return [self.F.sub(x[0], y[0]), self.F.sub(x[1], y[1])]

    def mul(self, x, y):
        F = self.F
        return [F.add(F.mul(x[0], y[0]), F.mul(self.W, F.mul(x[1], y[1]))), F.add(F.mul(x[0], y[1]), F.mul(x[1], y[0]))]

    def scalar(self, x, k):
        return [self.F.mul(x[0], k), self.F.mul(x[1], k)]


def _interpolation_gate(F, A, low_degree, subgroup_bits, w):
    """gates/high_degree_interpolation.rs:119-147 and gates/low_degree_interpolation.rs:356-404 (wire layout gates/interpolation.rs:19-76):
    shift at 0, the values at the 2^bits points from 1, evaluation point, evaluation value, coefficients; the low-degree gate appends the
    powers of the shift (i = 2..np-1, base wires) and of the evaluation point (i = 2..np-1)."""
    np_ = 1 << subgroup_bits
    out = []
    shift = w[0]
    start_values, eval_point, eval_value = 1, 1 + np_ * D, 1 + np_ * D + D
    start_coeffs = eval_value + D
    end_coeffs = start_coeffs + np_ * D
    coeffs = [A.get(w, start_coeffs + D * i) for i in range(np_)]
    g = pyref.root_of_unity(subgroup_bits)

    def eval_at_base(cs, x):  # PolynomialCoeffs<ext>::eval_base: Horner with a base-field point
        acc = [F.zero, F.zero]
        for c in reversed(cs):
            acc = A.add(A.scalar(acc, x), c)
        return acc

    if not low_degree:
        for i in range(np_):
            point = F.mul(F.c(pow(g, i, P)), shift)  # coset(shift) = g^i * shift
            out += A.sub(A.get(w, start_values + D * i), eval_at_base(coeffs, point))
        acc = [F.zero, F.zero]  # interpolant.eval(evaluation_point): Horner over the algebra
        ep = A.get(w, eval_point)
        for c in reversed(coeffs):
            acc = A.add(A.mul(acc, ep), c)
        out += A.sub(A.get(w, eval_value), acc)
        return out
    powers_shift = [F.one, shift] + [w[end_coeffs + i - 2] for i in range(2, np_)]
    for i in range(1, np_ - 1):
        out.append(F.sub(F.mul(powers_shift[i], shift), powers_shift[i + 1]))
    altered = [A.scalar(c, p) for c, p in zip(coeffs, powers_shift)]
    for i in range(np_):
        out += A.sub(A.get(w, start_values + D * i), eval_at_base(altered, F.c(pow(g, i, P))))
    epp = [None, A.get(w, eval_point)] + [A.get(w, end_coeffs + np_ - 2 + (i - 2) * D) for i in range(2, np_)]
    for i in range(1, np_ - 1):
        out += A.sub(A.mul(epp[i], epp[1]), epp[i + 1])
    acc = coeffs[0]  # eval_with_powers (field/src/polynomial/mod.rs:169-176): the ORIGINAL coefficients
    for i in range(1, np_):
        acc = A.add(acc, A.mul(epp[i], coeffs[i]))
    out += A.sub(A.get(w, eval_value), acc)
    return out


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
    A = Algebra(Base)
    rext = lambda: [rng.randrange(P), rng.randrange(P)]  # noqa: E731
    if kind == "arithmetic_extension":  # ArithmeticExtensionGenerator (arithmetic_extension.rs:216-262)
        w = []
        for _ in range(param):
            m0, m1, ad = rext(), rext(), rext()
            w += m0 + m1 + ad + A.add(A.scalar(A.mul(m0, m1), consts[0]), A.scalar(ad, consts[1]))
        return w
    if kind == "mul_extension":  # MulExtensionGenerator (multiplication_extension.rs:202-240)
        w = []
        for _ in range(param):
            m0, m1 = rext(), rext()
            w += m0 + m1 + A.scalar(A.mul(m0, m1), consts[0])
        return w
    if kind in ("reducing", "reducing_extension"):  # ReducingGenerator (reducing.rs:246-300, reducing_extension.rs:243-292)
        nc = param
        ext_coeffs = kind == "reducing_extension"
        alpha, acc = rext(), rext()
        coeffs = [rext() if ext_coeffs else [rng.randrange(P), 0] for _ in range(nc)]
        accs = []
        old = acc
        for c in coeffs:
            acc = A.add(A.mul(acc, alpha), c)
            accs.append(acc)
        flat_coeffs = [x for c in coeffs for x in (c if ext_coeffs else c[:1])]
        return accs[-1] + alpha + old + flat_coeffs + [x for a in accs[:-1] for x in a]
    if kind == "exponentiation":  # ExponentiationGenerator (exponentiation.rs:303-360)
        n = param
        base = rng.randrange(P)
        bits = [rng.randrange(2) for _ in range(n)]
        inter, cur = [], 1
        for i in range(n):
            prev = 1 if i == 0 else inter[i - 1] * inter[i - 1] % P
            cur = prev * (base if bits[n - 1 - i] else 1) % P
            inter.append(cur)
        return [base] + bits + [inter[-1]] + inter
    if kind == "poseidon_mds":  # PoseidonMdsGenerator (poseidon_mds.rs:253-300)
        inputs = [rext() for _ in range(SW)]
        outs = []
        for r in range(SW):
            acc = A.scalar(inputs[r], _C["POSEIDON_MDS_DIAG"][r])
            for i in range(SW):
                acc = A.add(acc, A.scalar(inputs[(i + r) % SW], _C["POSEIDON_MDS_CIRC"][i]))
            outs.append(acc)
        return [x for e in inputs + outs for x in e]
    if kind in ("high_degree_interpolation", "low_degree_interpolation"):  # InterpolationGenerator (both files)
        np_ = 1 << param
        g = pyref.root_of_unity(param)
        shift = rng.randrange(1, P)
        coeffs = [rext() for _ in range(np_)]

        def ev(x):  # x in the algebra
            acc = [0, 0]
            for c in reversed(coeffs):
                acc = A.add(A.mul(acc, x), c)
            return acc

        values = [ev([shift * pow(g, i, P) % P, 0]) for i in range(np_)]
        ep = rext()
        w = [shift] + [x for v in values for x in v] + ep + ev(ep) + [x for c in coeffs for x in c]
        if kind == "low_degree_interpolation":
            w += [pow(shift, i, P) for i in range(2, np_)]
            pw = ep
            for _ in range(2, np_):
                pw = A.mul(pw, ep)
                w += pw
        return w
    raise ValueError(kind)
