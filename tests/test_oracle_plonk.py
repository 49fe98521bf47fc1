"""Pins oracle/plonk_ref.py (partial products / Z, permutation terms of the vanishing polynomial,
compute_quotient_polys) with the verifier's identity vanishing(zeta) = Z_H(zeta) * t(zeta)."""
import random

import pytest

from oracle import plonk_ref, pyref
from plonk_instance import lde_leaves, make_circuit_instance, make_instance, poly_eval

P = pyref.P


@pytest.mark.parametrize("num_routed,degree_bits", [(10, 4), (17, 3), (16, 3)])
def test_quotient_satisfies_the_verifier_identity(num_routed, degree_bits):
    qdf, rate_bits = 8, 3
    inst = make_instance(degree_bits=degree_bits, num_wires=num_routed + 2, num_routed=num_routed, seed=num_routed)
    n, k_is = inst["n"], inst["k_is"]
    zpp = plonk_ref.zs_partial_products(inst["wires"], inst["sigmas"], k_is, inst["betas"], inst["gammas"], qdf, inst["subgroup"])
    num_prods = plonk_ref.num_partial_products(num_routed, qdf)
    assert len(zpp) == 2 * (1 + num_prods)
    # Z(1) = 1 and the grand product closes: Z(g^(n-1)) * (last row's total quotient) = 1
    for c in range(2):
        assert zpp[c][0] == 1
    w_c, w_l = lde_leaves(inst["wires"], rate_bits)
    cs_c, cs_l = lde_leaves(inst["constants"] + inst["sigmas"], rate_bits)
    z_c, z_l = lde_leaves(zpp, rate_bits)
    t = plonk_ref.compute_quotient_polys(w_l, cs_l, z_l, inst["num_constants"], k_is, inst["betas"], inst["gammas"], inst["alphas"],
                                         degree_bits, rate_bits, qdf)
    assert len(t) == 2 and len(t[0]) == n * 8
    # quotient degree: deg(prev_acc * prod of 8 numerators) - n < 8n - 8
    for c in range(2):
        assert all(v == 0 for v in t[c][8 * n - 8:])
    rng = random.Random(5)
    g = pyref.root_of_unity(degree_bits)
    for _ in range(3):
        zeta = rng.randrange(P)
        wires_z = [poly_eval(c, zeta) for c in w_c]
        sig_z = [poly_eval(c, zeta) for c in cs_c[inst["num_constants"]:]]
        zs_z = [poly_eval(z_c[c], zeta) for c in range(2)]
        zs_gz = [poly_eval(z_c[c], g * zeta % P) for c in range(2)]
        pp_z = [poly_eval(c, zeta) for c in z_c[2:]]
        zh = (pow(zeta, n, P) - 1) % P
        l0 = zh * plonk_ref.inv(n * (zeta - 1)) % P
        terms = plonk_ref.vanishing_terms_at(zeta, l0, wires_z, sig_z, zs_z, zs_gz, pp_z, k_is, inst["betas"], inst["gammas"], qdf, [])
        red = plonk_ref.reduce_with_powers_multi(terms, inst["alphas"])
        for c in range(2):
            assert red[c] == zh * poly_eval(t[c], zeta) % P


def test_broken_copy_constraint_is_not_divisible():
    """With one wire changed the vanishing polynomial is no longer a multiple of Z_H: the
    'quotient' interpolated on the coset fails the identity (so the test above is not vacuous)."""
    qdf, rate_bits, degree_bits, num_routed = 8, 3, 3, 10
    inst = make_instance(degree_bits=degree_bits, num_wires=12, num_routed=num_routed, seed=3, valid=False)
    zpp = plonk_ref.zs_partial_products(inst["wires"], inst["sigmas"], inst["k_is"], inst["betas"], inst["gammas"], qdf, inst["subgroup"])
    w_c, w_l = lde_leaves(inst["wires"], rate_bits)
    cs_c, cs_l = lde_leaves(inst["constants"] + inst["sigmas"], rate_bits)
    z_c, z_l = lde_leaves(zpp, rate_bits)
    t = plonk_ref.compute_quotient_polys(w_l, cs_l, z_l, 2, inst["k_is"], inst["betas"], inst["gammas"], inst["alphas"], degree_bits,
                                         rate_bits, qdf)
    assert any(v != 0 for v in t[0][8 * inst["n"] - 8:])


def test_gate_terms_enter_the_alpha_reduction_last():
    """Term order [L_0(Z-1)] | [partial-product checks] | [gate constraints] (vanishing_poly.rs:215-219)."""
    terms = plonk_ref.vanishing_terms_at(5, 7, [1, 2, 3], [4, 5, 6], [9], [10], [], [1, 7, 49], [11], [12], 8, [100, 200])
    assert terms[-2:] == [100, 200] and len(terms) == 1 + 1 + 2
    a = 3
    assert plonk_ref.reduce_with_powers_multi(terms, [a]) == [sum(t * pow(a, k, P) for k, t in enumerate(terms)) % P]


def test_ext2_evaluation_against_power_sums():
    """Horner in F_p[X]/(X^2-7) == sum of c_i * z^i with z^i from repeated multiplication; a base
    point (z1 = 0) reduces to the base-field evaluation; X*X = 7."""
    rng = random.Random(2)
    assert plonk_ref.ext2_mul((0, 1), (0, 1)) == (7, 0)
    for n in (1, 2, 5, 64):
        coeffs = [rng.randrange(P) for _ in range(n)]
        z = (rng.randrange(P), rng.randrange(P))
        acc, pw = (0, 0), (1, 0)
        for c in coeffs:
            acc = plonk_ref.ext2_add(acc, plonk_ref.ext2_mul((c, 0), pw))
            pw = plonk_ref.ext2_mul(pw, z)
        assert plonk_ref.eval_ext2(coeffs, z) == acc
        zb = (rng.randrange(P), 0)
        assert plonk_ref.eval_ext2(coeffs, zb) == (poly_eval(coeffs, zb[0]), 0)


def test_quadratic_extension_reference_constants():
    """The reference's own known answers for F_p[X]/(X^2 - 7) (field/src/goldilocks_extensions.rs:19-27 with the checks of
    field/src/field_testing.rs:154-166, test_power_of_two_gen): the multiplicative generator raised to (p^2 - 1) / 2^33 is the
    power-of-two generator, whose square is the base field's power-of-two generator (goldilocks_field.rs:89), and
    DTH_ROOT = W^((p-1)/2). They pin the extension multiplication every opening, FRI fold and gate evaluation of the oracle
    uses (plonk_ref.ext2_mul; gates_ref.Ext is checked against it)."""
    from oracle import fri_ref, gates_ref

    W, dth_root = 7, 18446744069414584320
    gen = (18081566051660590251, 16121475356294670766)
    pow2_gen = (0, 15659105665374529263)
    base_pow2_gen = 1753635133440165772
    assert plonk_ref.W == W and gates_ref.W == W
    assert pow(W, (P - 1) // 2, P) == dth_root == P - 1
    assert (P * P) >> 33 == (P * P - 1) >> 33  # order() >> TWO_ADICITY as the reference writes it
    assert fri_ref.ext_pow(gen, (P * P) >> 33) == pow2_gen
    assert fri_ref.ext_pow(pow2_gen, 2) == (base_pow2_gen, 0)
    assert fri_ref.ext_pow(pow2_gen, 1 << 33) == (1, 0) and fri_ref.ext_pow(pow2_gen, 1 << 32) != (1, 0)
    assert fri_ref.ext_mul(gen, fri_ref.ext_inv(gen)) == (1, 0)
    # Frobenius: x^p = (a, DTH_ROOT * b)  (field_testing.rs:135-144)
    assert fri_ref.ext_pow(gen, P) == (gen[0], dth_root * gen[1] % P)
    # the second extension class the oracle carries agrees
    assert gates_ref.Ext.mul(gen, pow2_gen) == fri_ref.ext_mul(gen, pow2_gen)
    assert gates_ref.Ext.mul(pow2_gen, pow2_gen) == (base_pow2_gen, 0)


@pytest.mark.parametrize("two_groups", [False, True])
def test_mini_circuit_with_gates_satisfies_the_verifier_identity(two_groups):
    """Gate constraints (filters from selector polynomials) + permutation argument on a tiny real
    circuit: the quotient built from the LDE equals vanishing(zeta) / Z_H(zeta) at random zeta."""
    qdf, rate_bits, degree_bits = 8, 3, 4
    inst = make_circuit_instance(degree_bits=degree_bits, seed=11 + two_groups, two_groups=two_groups)
    n, k_is, nc = inst["n"], inst["k_is"], inst["num_constants"]
    # the witness satisfies every gate on every row
    for r in range(n):
        lc = [col[r] for col in inst["constants"]]
        lw = [col[r] for col in inst["wires"]]
        assert plonk_ref.evaluate_gate_constraints(inst["gates"], inst["selector_indices"], inst["groups"], 4, lc, lw, inst["pih"]) == [0] * 4
    zpp = plonk_ref.zs_partial_products(inst["wires"], inst["sigmas"], k_is, inst["betas"], inst["gammas"], qdf, inst["subgroup"])
    w_c, w_l = lde_leaves(inst["wires"], rate_bits)
    cs_c, cs_l = lde_leaves(inst["constants"] + inst["sigmas"], rate_bits)
    z_c, z_l = lde_leaves(zpp, rate_bits)
    bits = degree_bits + rate_bits
    gate_terms = []
    for i in range(n * 8):
        leaf = pyref.reverse_bits(i, bits)
        gate_terms.append(plonk_ref.evaluate_gate_constraints(inst["gates"], inst["selector_indices"], inst["groups"], 4,
                                                              cs_l[leaf][:nc], w_l[leaf], inst["pih"]))
    t = plonk_ref.compute_quotient_polys(w_l, cs_l, z_l, nc, k_is, inst["betas"], inst["gammas"], inst["alphas"], degree_bits,
                                         rate_bits, qdf, gate_terms)
    rng = random.Random(6)
    g = pyref.root_of_unity(degree_bits)
    for _ in range(2):
        zeta = rng.randrange(P)
        wires_z = [poly_eval(c, zeta) for c in w_c]
        consts_z = [poly_eval(c, zeta) for c in cs_c[:nc]]
        gt = plonk_ref.evaluate_gate_constraints(inst["gates"], inst["selector_indices"], inst["groups"], 4, consts_z, wires_z, inst["pih"])
        zh = (pow(zeta, n, P) - 1) % P
        l0 = zh * plonk_ref.inv(n * (zeta - 1)) % P
        terms = plonk_ref.vanishing_terms_at(zeta, l0, wires_z, [poly_eval(c, zeta) for c in cs_c[nc:]],
                                             [poly_eval(z_c[c], zeta) for c in range(2)],
                                             [poly_eval(z_c[c], g * zeta % P) for c in range(2)],
                                             [poly_eval(c, zeta) for c in z_c[2:]], k_is, inst["betas"], inst["gammas"], qdf, gt)
        red = plonk_ref.reduce_with_powers_multi(terms, inst["alphas"])
        for c in range(2):
            assert red[c] == zh * poly_eval(t[c], zeta) % P


def test_reference_known_answer_for_partial_products():
    """The reference's one known answer for this stage (plonky2/src/util/partial_products.rs:114-142, test_partial_products):
    v = [1..6], denominators 1, Z(x) = 1, Z(gx) = 720."""
    v, ones = [1, 2, 3, 4, 5, 6], [1] * 6
    q2 = plonk_ref.quotient_chunk_products(v, 2)
    assert q2 == [2, 12, 30]
    pz2 = plonk_ref.partial_products_and_z_gx(1, q2)
    assert pz2 == [2, 24, 720]
    assert len(pz2) - 1 == plonk_ref.num_partial_products(len(v), 2) == 2
    assert all(t == 0 for t in plonk_ref.check_partial_products(v, ones, pz2[:-1], 1, 720, 2))
    q3 = plonk_ref.quotient_chunk_products(v, 3)
    assert q3 == [6, 120]
    pz3 = plonk_ref.partial_products_and_z_gx(1, q3)
    assert pz3 == [6, 720]
    assert len(pz3) - 1 == plonk_ref.num_partial_products(len(v), 3) == 1
    assert all(t == 0 for t in plonk_ref.check_partial_products(v, ones, pz3[:-1], 1, 720, 3))
    # and a wrong accumulator is caught
    assert any(t != 0 for t in plonk_ref.check_partial_products(v, ones, [2, 25], 1, 720, 2))
