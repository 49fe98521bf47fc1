"""plonk_ref.py — CPU restatement of the permutation-argument part of the prover hot path
(TEST INFRASTRUCTURE ONLY): partial products / Z, the non-gate terms of the vanishing polynomial,
and compute_quotient_polys with the gate-constraint terms supplied from outside.

Pure Python big ints, small sizes only. Each function cites the reference lines it restates.
Parity status: the reference holds no known answers for this stage, so besides
restating the code it is pinned by the verifier's identity (tests/test_oracle_plonk.py): for a
valid copy-constraint instance the quotient returned here satisfies
    vanishing(zeta) == Z_H(zeta) * t(zeta)
at random zeta, with every polynomial evaluated from its coefficients (independent of the LDE /
leaf indexing used to build t).
"""
from . import pyref

P = pyref.P


def inv(x):
    return pow(x % P, P - 2, P)


def quotient_chunk_products(quotient_values, max_degree):
    """plonky2/src/util/partial_products.rs:13-24"""
    out = []
    for k in range(0, len(quotient_values), max_degree):
        prod = 1
        for v in quotient_values[k : k + max_degree]:
            prod = prod * v % P
        out.append(prod)
    return out


def partial_products_and_z_gx(z_x, chunk_products):
    """partial_products.rs:28-37: running products of the chunk quotients starting from Z(x); the last one is Z(gx)."""
    acc, out = z_x % P, []
    for c in chunk_products:
        acc = acc * c % P
        out.append(acc)
    return out


def num_partial_products(n, max_degree):
    """partial_products.rs:41-48"""
    return -(-n // max_degree) - 1


def wires_permutation_partial_products_and_zs(wires, sigmas, k_is, beta, gamma, degree, subgroup):
    """plonky2/src/plonk/prover.rs:729-786. wires[j][i], sigmas[j][i] are columns (j < num_routed).
    Returns columns [pp_0 .. pp_{np-1}, Z] (Z last, as the reference returns them)."""
    num_routed, n = len(k_is), len(subgroup)
    num_prods = num_partial_products(num_routed, degree)
    rows = []
    z_x = 1
    for i, x in enumerate(subgroup):
        q = []
        for j in range(num_routed):
            w = wires[j][i]
            num = (w + beta * (k_is[j] * x % P) + gamma) % P
            den = (w + beta * sigmas[j][i] + gamma) % P
            q.append(num * inv(den) % P)
        chunks = quotient_chunk_products(q, degree)
        row = partial_products_and_z_gx(z_x, chunks)
        z_x, row[num_prods] = row[num_prods], z_x  # prover.rs:777-778: the last slot holds Z(x), not Z(gx)
        rows.append(row)
    return [[rows[i][k] for i in range(n)] for k in range(num_prods + 1)]


def zs_partial_products(wires, sigmas, k_is, betas, gammas, degree, subgroup):
    """prover.rs:106-117: Z of every challenge first, then the partial products challenge-major."""
    per = [wires_permutation_partial_products_and_zs(wires, sigmas, k_is, b, g, degree, subgroup) for b, g in zip(betas, gammas)]
    zs = [p[-1] for p in per]
    pps = [col for p in per for col in p[:-1]]
    return zs + pps


def check_partial_products(numerators, denominators, partials, z_x, z_gx, max_degree):
    """partial_products.rs:52-76"""
    accs = [z_x] + list(partials) + [z_gx]
    out = []
    for c, k in enumerate(range(0, len(numerators), max_degree)):
        np_, dp = 1, 1
        for v in numerators[k : k + max_degree]:
            np_ = np_ * v % P
        for v in denominators[k : k + max_degree]:
            dp = dp * v % P
        out.append((accs[c] * np_ - accs[c + 1] * dp) % P)
    return out


def reduce_with_powers_multi(terms, alphas):
    """plonk_common.rs:97-114: Horner from the last term, cumul = term + cumul * alpha."""
    cumul = [0] * len(alphas)
    for t in reversed(terms):
        cumul = [(t + c * a) % P for c, a in zip(cumul, alphas)]
    return cumul


def vanishing_terms_at(x, l_0_x, local_wires, s_sigmas, local_zs, next_zs, partial_products, k_is, betas, gammas,
                       degree, gate_terms):
    """vanishing_poly.rs:146-221 for one point: [L_0(x)(Z_i - 1)] + [partial-product checks] + [gate terms]."""
    num_routed = len(k_is)
    num_prods = num_partial_products(num_routed, degree)
    z1, pp = [], []
    for i, (beta, gamma) in enumerate(zip(betas, gammas)):
        z_x, z_gx = local_zs[i], next_zs[i]
        z1.append(l_0_x * (z_x - 1) % P)
        nums = [(local_wires[j] + beta * (k_is[j] * x % P) + gamma) % P for j in range(num_routed)]
        dens = [(local_wires[j] + beta * s_sigmas[j] + gamma) % P for j in range(num_routed)]
        pp += check_partial_products(nums, dens, partial_products[i * num_prods : (i + 1) * num_prods], z_x, z_gx, degree)
    return z1 + pp + list(gate_terms)


def compute_quotient_polys(wires_leaves, cs_leaves, zpp_leaves, num_constants, k_is, betas, gammas, alphas, degree_bits,
                           rate_bits, quotient_degree_factor, gate_terms=None, shift=pyref.GENERATOR):
    """plonky2/src/plonk/prover.rs:790-1034 with the gate-constraint terms given per LDE point
    (gate_terms[i] = list, or None for a circuit without gate constraints).
    *_leaves are the commitments' leaf-major LDE rows (leaf j = evaluations at bitrev(j)), as
    PolynomialBatch::get_lde_values reads them (fri/oracle.rs:1007-1018).
    Returns num_challenges coefficient vectors of length n << quotient_degree_bits."""
    qdb = (quotient_degree_factor - 1).bit_length()  # log2_ceil
    assert qdb <= rate_bits
    step, next_step = 1 << (rate_bits - qdb), 1 << qdb
    n = 1 << degree_bits
    lde_size = n << qdb
    num_routed, num_ch = len(k_is), len(betas)
    w = pyref.root_of_unity(degree_bits + qdb)
    # ZeroPolyOnCoset::new(degree_bits, qdb), field/src/zero_poly_coset.rs:20-33
    g_pow_n = pow(shift, n, P)
    zh_evals = [(g_pow_n * pow(pyref.root_of_unity(qdb), k, P) - 1) % P for k in range(1 << qdb)]
    bits = degree_bits + rate_bits

    def lde_row(leaves, i):
        return leaves[pyref.reverse_bits(i * step, bits)]

    out = [[0] * lde_size for _ in range(num_ch)]
    xw = 1
    for i in range(lde_size):
        x = shift * xw % P  # shifted_x, prover.rs:903
        xw = xw * w % P
        i_next = (i + next_step) % lde_size
        cs = lde_row(cs_leaves, i)
        s_sigmas = cs[num_constants : num_constants + num_routed]
        local_wires = lde_row(wires_leaves, i)
        zpp = lde_row(zpp_leaves, i)
        local_zs, partial_products = zpp[:num_ch], zpp[num_ch:]
        next_zs = lde_row(zpp_leaves, i_next)[:num_ch]
        zh = zh_evals[i % (1 << qdb)]
        l_0_x = zh * inv(n * (x - 1)) % P  # eval_l_0, zero_poly_coset.rs:57-60
        terms = vanishing_terms_at(x, l_0_x, local_wires, s_sigmas, local_zs, next_zs, partial_products, k_is, betas, gammas,
                                   quotient_degree_factor, gate_terms[i] if gate_terms is not None else [])
        red = reduce_with_powers_multi(terms, alphas)
        zh_inv = inv(zh)
        for c in range(num_ch):
            out[c][i] = red[c] * zh_inv % P  # prover.rs:985-991
    return [pyref.coset_idft_fast(col, shift) for col in out]  # prover.rs:1009-1021


# ---------------------------------------------------------------- quadratic extension, openings
W = 7  # Extendable<2>::W, field/src/goldilocks_extensions.rs:19


def ext2_mul(x, y):
    """field/src/extension/quadratic.rs:173-185: (a0 + a1 X)(b0 + b1 X) with X^2 = W."""
    a0, a1 = x
    b0, b1 = y
    return ((a0 * b0 + W * a1 * b1) % P, (a0 * b1 + a1 * b0) % P)


def ext2_add(x, y):
    return ((x[0] + y[0]) % P, (x[1] + y[1]) % P)


def eval_ext2(coeffs, z):
    """p.to_extension().eval(z): Horner from the top coefficient (field/src/polynomial/mod.rs:161-166)."""
    acc = (0, 0)
    for c in reversed(coeffs):
        acc = ext2_add(ext2_mul(acc, z), (c % P, 0))
    return acc


def opening_evals(polynomials, zeta, g=None):
    """eval_commitment of OpeningSet::new (plonky2/src/plonk/proof.rs:314-333) at zeta and,
    when g is given, at g*zeta (plonk_zs_next)."""
    pts = [zeta] + ([ext2_mul((g % P, 0), zeta)] if g is not None else [])
    return [[eval_ext2(p, z) for p in polynomials] for z in pts]


# ---------------------------------------------------------------- gate constraints
UNUSED_SELECTOR = (1 << 32) - 1  # plonky2/src/gates/selectors.rs:11


def compute_filter(row, group_range, s, many_selectors):
    """plonky2/src/gates/gate.rs:261-268"""
    f = 1
    for i in list(range(*group_range)) + ([UNUSED_SELECTOR] if many_selectors else []):
        if i != row:
            f = f * ((i - s) % P) % P
    return f


# unfiltered constraints of the gates used by the tests; vars = (local_constants after the selector
# prefix, local_wires, public_inputs_hash)
def arithmetic_gate(num_ops):
    """ArithmeticGate::eval_unfiltered_base_packed (plonky2/src/gates/arithmetic_base.rs:199-216)"""
    def f(consts, wires, pih):
        c0, c1 = consts[0], consts[1]
        return [(wires[4 * i + 3] - (wires[4 * i] * wires[4 * i + 1] % P * c0 + wires[4 * i + 2] * c1)) % P for i in range(num_ops)]
    return f


def constant_gate(num_consts):
    """ConstantGate (plonky2/src/gates/constant.rs:150-158)"""
    return lambda consts, wires, pih: [(consts[i] - wires[i]) % P for i in range(num_consts)]


def public_input_gate():
    """PublicInputGate (plonky2/src/gates/public_input.rs:129-139)"""
    return lambda consts, wires, pih: [(wires[i] - pih[i]) % P for i in range(4)]


def noop_gate():
    """NoopGate (plonky2/src/gates/noop.rs): no constraints"""
    return lambda consts, wires, pih: []


def evaluate_gate_constraints(gates, selector_indices, groups, num_gate_constraints, local_constants, local_wires, pih):
    """evaluate_gate_constraints_base_batch for one point (plonky2/src/plonk/vanishing_poly.rs:267-306)
    over Gate::eval_filtered (gates/gate.rs:86-109): constraints[k] += filter_g * c_{g,k}."""
    num_selectors = len(groups)
    out = [0] * num_gate_constraints
    for row, gate in enumerate(gates):
        si = selector_indices[row]
        filt = compute_filter(row, groups[si], local_constants[si], num_selectors > 1)
        for k, c in enumerate(gate(local_constants[num_selectors:], local_wires, pih)):
            out[k] = (out[k] + filt * c) % P
    return out
