"""HIP partial products / Z and quotient polynomials vs oracle/plonk_ref.py (itself pinned by the
verifier identity in test_oracle_plonk.py). Bit-exact."""
import random

import numpy as np
import pytest

from gpu_util import gpu  # noqa: F401
from oracle import plonk_ref, pyref
from plonk_instance import lde_leaves, make_circuit_instance, make_instance, poly_eval

pytestmark = pytest.mark.gpu
P = pyref.P


def cols(a):
    return np.array(a, dtype=np.uint64)


@pytest.mark.parametrize("num_routed,degree_bits,qdf,num_ch", [(10, 4, 8, 2), (17, 3, 8, 2), (16, 5, 8, 1), (80, 6, 8, 2),
                                                               (9, 11, 4, 3), (12, 13, 8, 2)])
def test_partial_products_and_zs(gpu, num_routed, degree_bits, qdf, num_ch):
    import plonky2_gpu_amd as pg

    inst = make_instance(degree_bits=degree_bits, num_wires=num_routed + 3, num_routed=num_routed, num_challenges=num_ch,
                         seed=num_routed * 7 + degree_bits, valid=(degree_bits <= 6))
    n = inst["n"]
    if degree_bits > 8:  # big sizes: random (not copy-consistent) data is enough for kernel-vs-oracle parity
        rng = np.random.default_rng(degree_bits)
        for key in ("wires", "sigmas"):
            inst[key] = [[int(v) % P for v in rng.integers(0, 2**63, size=n)] for _ in inst[key]]
    exp = plonk_ref.zs_partial_products(inst["wires"], inst["sigmas"], inst["k_is"], inst["betas"], inst["gammas"], qdf,
                                        inst["subgroup"])
    d_w = pg.DeviceBuffer.from_host(gpu, cols(inst["wires"]))
    d_s = pg.DeviceBuffer.from_host(gpu, cols(inst["sigmas"]))
    d_k = pg.DeviceBuffer.from_host(gpu, cols(inst["k_is"]))
    out, n_cols = pg.all_wires_permutation_partial_products(gpu, d_w, n, d_s, n, d_k, inst["betas"], inst["gammas"], num_routed,
                                                            qdf, degree_bits)
    got = out.download().reshape(n_cols, n)
    assert n_cols == len(exp)
    assert (got == cols(exp)).all()


@pytest.mark.parametrize("qdf,expected_pps", [(2, [2, 24]), (3, [6])])
def test_partial_products_reference_known_answer(gpu, qdf, expected_pps):
    """plonky2/src/util/partial_products.rs:114-142 on the device: wires and sigmas chosen so that every row's
    numerators are [1..6] and its denominators 1 (beta = 1, gamma = 0: w_j = v_j - k_j x, sigma_j = 1 - w_j). Row 0 must
    hold the reference's partial products ([2, 24] for degree 2, [6] for degree 3) and Z(g x_0) = 720; every later row the
    same values times 720^row."""
    import plonky2_gpu_amd as pg

    degree_bits, num_routed = 3, 6
    n = 1 << degree_bits
    w = pow(7, (P - 1) >> degree_bits, P)
    subgroup = [pow(w, i, P) for i in range(n)]
    k_is = [pow(7, j, P) for j in range(num_routed)]
    v = [1, 2, 3, 4, 5, 6]
    wires = [[(v[j] - k_is[j] * x) % P for x in subgroup] for j in range(num_routed)]
    sigmas = [[(1 - wires[j][i]) % P for i in range(n)] for j in range(num_routed)]
    exp = plonk_ref.zs_partial_products(wires, sigmas, k_is, [1], [0], qdf, subgroup)
    d_w = pg.DeviceBuffer.from_host(gpu, cols(wires))
    d_s = pg.DeviceBuffer.from_host(gpu, cols(sigmas))
    d_k = pg.DeviceBuffer.from_host(gpu, cols(k_is))
    out, n_cols = pg.all_wires_permutation_partial_products(gpu, d_w, n, d_s, n, d_k, [1], [0], num_routed, qdf, degree_bits)
    got = out.download().reshape(n_cols, n)
    assert n_cols == 1 + len(expected_pps)
    z, pps = got[0], got[1:]
    assert int(z[0]) == 1 and int(z[1]) == 720
    assert [int(c[0]) for c in pps] == expected_pps
    for i in range(n):
        assert int(z[i]) == pow(720, i, P)
        assert [int(c[i]) for c in pps] == [e * pow(720, i, P) % P for e in expected_pps]
    assert (got == cols(exp)).all()


def test_partial_products_argument_errors(gpu):
    import plonky2_gpu_amd as pg

    buf = pg.DeviceBuffer(gpu, 1024)
    with pytest.raises(pg.Plonky2HipError):  # prover.rs:102-105: degree must be < num_routed_wires
        pg.all_wires_permutation_partial_products(gpu, buf, 16, buf, 16, buf, [1], [2], 8, 8, 4)
    with pytest.raises(pg.Plonky2HipError):
        pg.all_wires_permutation_partial_products(gpu, buf, 16, buf, 16, buf, [1] * 5, [2] * 5, 10, 8, 4)


@pytest.mark.parametrize("num_routed,degree_bits,qdf,with_gates", [(10, 4, 8, False), (17, 3, 8, True), (12, 5, 4, True),
                                                                   (80, 6, 8, False)])
def test_compute_quotient_polys(gpu, oracle, num_routed, degree_bits, qdf, with_gates):
    """End to end on the device: commit wires / constants+sigmas, partial products + Z, commit
    them, quotient polynomials — equal to the oracle's, and (valid instance, no gates) satisfying
    the verifier identity at a random point."""
    import plonky2_gpu_amd as pg

    rate_bits, cap_h, num_constants = 3, 2, 2
    inst = make_instance(degree_bits=degree_bits, num_wires=num_routed + 2, num_routed=num_routed, num_constants=num_constants,
                         seed=100 + num_routed)
    n, k_is = inst["n"], inst["k_is"]
    wires_b = pg.PolynomialBatch.from_values(gpu, cols(inst["wires"]), rate_bits, False, cap_h)
    cs_b = pg.PolynomialBatch.from_values(gpu, cols(inst["constants"] + inst["sigmas"]), rate_bits, False, cap_h)
    d_w = pg.DeviceBuffer.from_host(gpu, cols(inst["wires"]))
    d_s = pg.DeviceBuffer.from_host(gpu, cols(inst["sigmas"]))
    d_k = pg.DeviceBuffer.from_host(gpu, cols(k_is))
    d_zpp, n_cols = pg.all_wires_permutation_partial_products(gpu, d_w, n, d_s, n, d_k, inst["betas"], inst["gammas"], num_routed,
                                                              qdf, degree_bits)
    zpp_host = d_zpp.download().reshape(n_cols, n)
    zpp_b = pg.PolynomialBatch.from_values_device(gpu, d_zpp, n_cols, degree_bits, rate_bits, False, cap_h)
    qdb = (qdf - 1).bit_length()
    lde_size = n << qdb
    gate_terms, d_gt, ngc = None, None, 0
    if with_gates:
        ngc = 5
        rng = random.Random(9)
        gate_terms = [[rng.randrange(P) for _ in range(ngc)] for _ in range(lde_size)]
        d_gt = pg.DeviceBuffer.from_host(gpu, cols(gate_terms))
    d_q = pg.compute_quotient_polys(gpu, wires_b, cs_b, zpp_b, num_constants, num_routed, d_k, inst["betas"], inst["gammas"],
                                    inst["alphas"], qdf, d_gt, ngc)
    got = d_q.download().reshape(2, lde_size)
    # oracle on the same leaves (taken from the oracle's own commit of the same values)
    w_c, w_l = lde_leaves(inst["wires"], rate_bits)
    cs_c, cs_l = lde_leaves(inst["constants"] + inst["sigmas"], rate_bits)
    z_c, z_l = lde_leaves(zpp_host.tolist(), rate_bits)
    exp = plonk_ref.compute_quotient_polys(w_l, cs_l, z_l, num_constants, k_is, inst["betas"], inst["gammas"], inst["alphas"],
                                           degree_bits, rate_bits, qdf, gate_terms)
    assert (got == cols(exp)).all()
    if not with_gates and qdf == 8:
        zeta = 0x1234567890ABCDEF % P
        g = pyref.root_of_unity(degree_bits)
        zh = (pow(zeta, n, P) - 1) % P
        l0 = zh * plonk_ref.inv(n * (zeta - 1)) % P
        terms = plonk_ref.vanishing_terms_at(
            zeta, l0, [poly_eval(c, zeta) for c in w_c], [poly_eval(c, zeta) for c in cs_c[num_constants:]],
            [poly_eval(z_c[c], zeta) for c in range(2)], [poly_eval(z_c[c], g * zeta % P) for c in range(2)],
            [poly_eval(c, zeta) for c in z_c[2:]], k_is, inst["betas"], inst["gammas"], qdf, [])
        red = plonk_ref.reduce_with_powers_multi(terms, inst["alphas"])
        for c in range(2):
            assert red[c] == zh * poly_eval([int(v) for v in got[c]], zeta) % P


@pytest.mark.parametrize("n_polys,log_n", [(3, 0), (5, 3), (7, 8), (20, 12), (135, 10), (4, 16), (2, 19)])
def test_opening_evaluations_in_the_quadratic_extension(gpu, oracle, n_polys, log_n):
    """OpeningSet::new's eval_commitment (plonk/proof.rs:314-333): every polynomial of a commitment
    at zeta and g*zeta in F_{p^2}, computed where the coefficients live."""
    import plonky2_gpu_amd as pg

    n = 1 << log_n
    coeffs = oracle.random_field((n_polys, n), seed=n_polys * 31 + log_n)
    batch = pg.PolynomialBatch.from_coeffs(gpu, coeffs, 1, False, 0, leaf_major=False)
    rng = random.Random(log_n)
    zeta = (rng.randrange(P), rng.randrange(P))
    g = pyref.root_of_unity(max(log_n, 1))
    pts = [zeta, plonk_ref.ext2_mul((g, 0), zeta), (5, 0)]
    got = batch.eval_polynomials_ext2(pts)
    sample = range(n_polys) if n <= 4096 else [0, n_polys - 1]
    for q, z in enumerate(pts):
        for i in sample:
            assert tuple(int(v) for v in got[q, i]) == plonk_ref.eval_ext2([int(c) for c in coeffs[i]], z), (q, i)


def test_quadratic_extension_reference_constants_on_the_device(gpu):
    """The device's F_{p^2} arithmetic against the reference's constants (field/src/goldilocks_extensions.rs:24-27,
    field/src/field_testing.rs:154-166): evaluating X^k at the extension's power-of-two generator g = (0, 15659105665374529263)
    gives g^k — g^2 is the base field's power-of-two generator 1753635133440165772, g^(2^15) has order 2^18 (its 2^18-th power
    through another evaluation is one) — and at the multiplicative generator the evaluation of X^(2^16) equals the oracle's
    power."""
    import plonky2_gpu_amd as pg
    from oracle import fri_ref

    gen = (18081566051660590251, 16121475356294670766)
    g = (0, 15659105665374529263)
    log_n = 17
    coeffs = np.zeros((4, 1 << log_n), dtype=np.uint64)
    coeffs[0, 2] = 1  # X^2
    coeffs[1, 1 << 15] = 1  # X^(2^15)
    coeffs[2, 1 << 16] = 1  # X^(2^16)
    coeffs[3, :3] = (5, 0, 3)  # 5 + 3 X^2
    batch = pg.PolynomialBatch.from_coeffs(gpu, coeffs, 1, False, 0, leaf_major=False)
    got = batch.eval_polynomials_ext2([g, gen])
    val = lambda q, i: tuple(int(v) for v in got[q, i])
    assert val(0, 0) == (1753635133440165772, 0)
    h = val(0, 1)
    assert h == fri_ref.ext_pow(g, 1 << 15) and fri_ref.ext_pow(h, 1 << 18) == (1, 0) and fri_ref.ext_pow(h, 1 << 17) != (1, 0)
    assert val(1, 2) == fri_ref.ext_pow(gen, 1 << 16)
    assert val(0, 3) == ((5 + 3 * 1753635133440165772) % P, 0)
    # the device value fed back: evaluating X^(2^16) at h = g^(2^15) is g^(2^31), whose fourth power is one
    again = batch.eval_polynomials_ext2([h])
    g31 = tuple(int(v) for v in again[0, 2])
    assert g31 == fri_ref.ext_pow(g, 1 << 31) and fri_ref.ext_pow(g31, 4) == (1, 0) and fri_ref.ext_pow(g31, 2) != (1, 0)


@pytest.mark.parametrize("two_groups,degree_bits", [(False, 4), (True, 4), (True, 7)])
def test_quotient_with_table_driven_gates(gpu, two_groups, degree_bits):
    """compute_quotient_polys for a circuit described by gate programs (Noop / Constant / PublicInput /
    Arithmetic): equals the oracle's evaluate_gate_constraints path and satisfies the verifier
    identity with the gate constraints included."""
    import plonky2_gpu_amd as pg
    from plonky2_gpu_amd import gate_program as gp

    qdf, rate_bits, cap_h = 8, 3, 2
    inst = make_circuit_instance(degree_bits=degree_bits, seed=21 + two_groups, two_groups=two_groups)
    n, k_is, nc = inst["n"], inst["k_is"], inst["num_constants"]
    wires_b = pg.PolynomialBatch.from_values(gpu, cols(inst["wires"]), rate_bits, False, cap_h)
    cs_b = pg.PolynomialBatch.from_values(gpu, cols(inst["constants"] + inst["sigmas"]), rate_bits, False, cap_h)
    d_w = pg.DeviceBuffer.from_host(gpu, cols(inst["wires"]))
    d_s = pg.DeviceBuffer.from_host(gpu, cols(inst["sigmas"]))
    d_k = pg.DeviceBuffer.from_host(gpu, cols(k_is))
    d_zpp, n_cols = pg.all_wires_permutation_partial_products(gpu, d_w, n, d_s, n, d_k, inst["betas"], inst["gammas"], 12, qdf,
                                                              degree_bits)
    zpp_host = d_zpp.download().reshape(n_cols, n)
    zpp_b = pg.PolynomialBatch.from_values_device(gpu, d_zpp, n_cols, degree_bits, rate_bits, False, cap_h)
    prog = pg.GateProgram(gpu, [gp.noop_gate(), gp.constant_gate(2), gp.public_input_gate(), gp.arithmetic_gate(3)],
                          inst["selector_indices"], inst["groups"], inst["pih"])
    d_q = pg.compute_quotient_polys(gpu, wires_b, cs_b, zpp_b, nc, 12, d_k, inst["betas"], inst["gammas"], inst["alphas"], qdf,
                                    None, inst["num_gate_constraints"], prog)
    got = d_q.download().reshape(2, n * 8)
    # the four ways of running the same thing agree: interpreter / run-time compiled gates x leaf-major / column-major
    alt = pg.compute_quotient_polys(gpu, wires_b, cs_b, zpp_b, nc, 12, d_k, inst["betas"], inst["gammas"], inst["alphas"], qdf,
                                    None, inst["num_gate_constraints"], prog, column_major=False)
    assert (alt.download().reshape(2, n * 8) == got).all()
    prog.compile(inst["num_gate_constraints"], 2)
    assert "gate_3" in prog.kernel_source() and "gl::mul" in prog.kernel_source()
    for cm in (True, False):
        alt = pg.compute_quotient_polys(gpu, wires_b, cs_b, zpp_b, nc, 12, d_k, inst["betas"], inst["gammas"], inst["alphas"], qdf,
                                        None, inst["num_gate_constraints"], prog, column_major=cm)
        assert (alt.download().reshape(2, n * 8) == got).all()
    w_c, w_l = lde_leaves(inst["wires"], rate_bits)
    cs_c, cs_l = lde_leaves(inst["constants"] + inst["sigmas"], rate_bits)
    z_c, z_l = lde_leaves(zpp_host.tolist(), rate_bits)
    bits = degree_bits + rate_bits
    if degree_bits <= 4:
        gate_terms = [plonk_ref.evaluate_gate_constraints(inst["gates"], inst["selector_indices"], inst["groups"], 4,
                                                          cs_l[pyref.reverse_bits(i, bits)][:nc], w_l[pyref.reverse_bits(i, bits)],
                                                          inst["pih"]) for i in range(n * 8)]
        exp = plonk_ref.compute_quotient_polys(w_l, cs_l, z_l, nc, k_is, inst["betas"], inst["gammas"], inst["alphas"], degree_bits,
                                               rate_bits, qdf, gate_terms)
        assert (got == cols(exp)).all()
    # verifier identity with gate constraints at a random point
    zeta = 0xFEDCBA9876543210 % P
    g = pyref.root_of_unity(degree_bits)
    wires_z = [poly_eval(c, zeta) for c in w_c]
    consts_z = [poly_eval(c, zeta) for c in cs_c[:nc]]
    gt = plonk_ref.evaluate_gate_constraints(inst["gates"], inst["selector_indices"], inst["groups"], 4, consts_z, wires_z, inst["pih"])
    zh = (pow(zeta, n, P) - 1) % P
    l0 = zh * plonk_ref.inv(n * (zeta - 1)) % P
    terms = plonk_ref.vanishing_terms_at(zeta, l0, wires_z, [poly_eval(c, zeta) for c in cs_c[nc:]],
                                         [poly_eval(z_c[c], zeta) for c in range(2)], [poly_eval(z_c[c], g * zeta % P) for c in range(2)],
                                         [poly_eval(c, zeta) for c in z_c[2:]], k_is, inst["betas"], inst["gammas"], qdf, gt)
    red = plonk_ref.reduce_with_powers_multi(terms, inst["alphas"])
    for c in range(2):
        assert red[c] == zh * poly_eval([int(v) for v in got[c]], zeta) % P
