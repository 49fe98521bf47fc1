"""prove_ref.py — CPU restatement of the whole prover and verifier around the hot path
(TEST INFRASTRUCTURE ONLY): prove() (plonky2/src/plonk/prover.rs:40-233), get_challenges
(plonk/get_challenges.rs:28-75) and verify_with_challenges (plonk/verifier.rs:39-120), assembled from
the stage restatements in pyref / plonk_ref / fri_ref. Pure Python big ints, small circuits only.

Parity status: the reference holds no proof fixtures, so this is pinned the way the
reference pins its own prover — `prove` then `verify` (plonk/prover.rs tests, examples/): the
verifier below recomputes every challenge from the proof alone, checks vanishing(zeta) ==
Z_H(zeta) * t(zeta) over the quadratic extension with the gate constraints evaluated independently
of the prover's base-field path, and runs the FRI verifier; tampering with any part is rejected
(tests/test_oracle_prove.py).

A circuit is a dict:
  degree_bits, num_wires, num_routed_wires, num_constants, num_challenges, quotient_degree_factor,
  k_is, gates = [(kind, param)], selector_indices, groups, num_gate_constraints,
  constants (columns, selectors first), sigmas (columns), fri_params, circuit_digest,
  constants_sigmas = the preprocessed commitment {"polynomials","leaves","digests","cap"}.
"""
from . import fri_ref, gates_ref, plonk_ref, pyref

P = pyref.P
ext_add, ext_sub, ext_mul, ext_inv, ext_pow = fri_ref.ext_add, fri_ref.ext_sub, fri_ref.ext_mul, fri_ref.ext_inv, fri_ref.ext_pow


def hash_pad(inputs):
    """Hasher::hash_pad (plonky2/src/plonk/config.rs:44-52): pad10*1 to a multiple of SPONGE_WIDTH."""
    padded = list(inputs) + [1]
    while (len(padded) + 1) % 12 != 0:
        padded.append(0)
    padded.append(1)
    return pyref.hash_no_pad(padded)


def circuit_digest(constants_sigmas_cap, degree_bits, domain_separator=()):
    """circuit_builder.rs:915-927"""
    parts = [x for h in constants_sigmas_cap for x in h] + hash_pad(domain_separator) + [degree_bits]
    return pyref.hash_no_pad(parts)


SALT_SIZE = 4  # fri/oracle.rs:41


def commit_from_coeffs(coeffs, rate_bits, cap_height, salt=None):
    """PolynomialBatch::from_coeffs (fri/oracle.rs:911-977). `salt` (blinding, oracle.rs:985-1002: SALT_SIZE extra columns of
    rate * n random elements appended to the LDE's columns before the transposition): SALT_SIZE columns given in LEAF order —
    entry j is the element of leaf j, i.e. of the natural-order random vector at bitrev(j); the reference draws them from OsRng,
    so any order of a uniform vector is the same distribution and the leaf order is what a device buffer holds."""
    n_ext = len(coeffs[0]) << rate_bits
    lde = []
    for c in coeffs:
        scaled = [x * pow(pyref.GENERATOR, i, P) % P for i, x in enumerate(c)] + [0] * (n_ext - len(c))
        lde.append(pyref.fast_ntt(scaled))
    lg = pyref.log2_strict(n_ext)
    leaves = [[col[pyref.reverse_bits(i, lg)] for col in lde] for i in range(n_ext)]
    if salt is not None:
        assert len(salt) == SALT_SIZE and all(len(col) == n_ext for col in salt)
        leaves = [row + [int(col[i]) % P for col in salt] for i, row in enumerate(leaves)]
    digests, cap = pyref.merkle_tree(leaves, cap_height)
    return dict(polynomials=[list(c) for c in coeffs], leaves=leaves, digests=digests, cap=cap)


def commit_from_values(values, rate_bits, cap_height, salt=None):
    if salt is not None:  # from_values = ifft per column, then from_coeffs (oracle.rs:709-731)
        return commit_from_coeffs([pyref.fast_ntt(list(v), inverse=True) for v in values], rate_bits, cap_height, salt)
    coeffs, leaves, digests, cap = pyref.commit_from_values(values, rate_bits, cap_height)
    return dict(polynomials=coeffs, leaves=leaves, digests=digests, cap=cap)


def base_gates(circuit):
    mk = dict(noop=lambda p: plonk_ref.noop_gate(), constant=plonk_ref.constant_gate, public_input=lambda p: plonk_ref.public_input_gate(),
              arithmetic=plonk_ref.arithmetic_gate)

    def other(kind, param):  # the rest of the ed25519 gate list lives in gates_ref.py
        return lambda consts, wires, pih: gates_ref.constraints(kind, param, consts, wires, pih, gates_ref.Base)

    return [mk[kind](param) if kind in mk else other(kind, param) for kind, param in circuit["gates"]]


def fri_instance(circuit, zeta):
    """CommonCircuitData::get_fri_instance (plonk/circuit_data.rs:351-371): every polynomial of the four
    oracles [constants_sigmas, wires, zs_partial_products, quotient] at zeta, the Zs also at g*zeta."""
    nc = circuit["num_challenges"]
    counts = [circuit["num_constants"] + circuit["num_routed_wires"], circuit["num_wires"],
              nc * (1 + plonk_ref.num_partial_products(circuit["num_routed_wires"], circuit["quotient_degree_factor"])),
              nc * circuit["quotient_degree_factor"]]
    all_polys = [(oi, pi) for oi, k in enumerate(counts) for pi in range(k)]
    g = pyref.root_of_unity(circuit["degree_bits"])
    return dict(batches=[(zeta, all_polys), (ext_mul((g, 0), zeta), [(2, i) for i in range(nc)])])


def fri_openings(openings):
    """OpeningSet::to_fri_openings (plonk/proof.rs:336-356)"""
    return [openings["constants"] + openings["plonk_sigmas"] + openings["wires"] + openings["plonk_zs"] + openings["partial_products"]
            + openings["quotient_polys"], openings["plonk_zs_next"]]


def prove(circuit, wires, public_inputs, trace=None, salts=None):
    """plonk/prover.rs:40-233 from the full witness (wire columns) on. `trace` (a dict) receives the intermediate objects
    the reference can dump (prover.rs:829-877): commitments, Z / partial-product values, challenges, quotient polynomials.
    With fri_params["hiding"] (CircuitConfig::zero_knowledge, circuit_data.rs:74) the wires, Zs / partial products and quotient
    commitments are blinded (prover.rs:84, 125, 174): `salts` = [3][SALT_SIZE][n_ext] in leaf order (see commit_from_coeffs)."""
    fp = circuit["fri_params"]
    hiding = bool(fp.get("hiding"))
    assert hiding == (salts is not None), "salts are given exactly when the circuit is hiding"
    salt_w, salt_z, salt_q = salts if hiding else (None, None, None)
    rate_bits, cap_height = fp["rate_bits"], fp["cap_height"]
    db, n = circuit["degree_bits"], 1 << circuit["degree_bits"]
    nch, qdf, num_routed = circuit["num_challenges"], circuit["quotient_degree_factor"], circuit["num_routed_wires"]
    pih = pyref.hash_no_pad(public_inputs)
    wires_c = commit_from_values(wires, rate_bits, cap_height, salt_w)
    ch = fri_ref.Challenger()
    ch.observe_elements(circuit["circuit_digest"])
    ch.observe_elements(pih)
    ch.observe_cap(wires_c["cap"])
    betas, gammas = ch.get_n_challenges(nch), ch.get_n_challenges(nch)
    assert qdf < num_routed
    subgroup = [pow(pyref.root_of_unity(db), i, P) for i in range(n)]
    zs_pp = plonk_ref.zs_partial_products(wires, circuit["sigmas"], circuit["k_is"], betas, gammas, qdf, subgroup)
    zs_c = commit_from_values(zs_pp, rate_bits, cap_height, salt_z)
    ch.observe_cap(zs_c["cap"])
    alphas = ch.get_n_challenges(nch)
    cs = circuit["constants_sigmas"]
    qdb = (qdf - 1).bit_length()
    bits, step = db + rate_bits, 1 << (rate_bits - qdb)
    gates = base_gates(circuit)
    gate_terms = []
    for i in range(n << qdb):
        row = pyref.reverse_bits(i * step, bits)
        gate_terms.append(plonk_ref.evaluate_gate_constraints(gates, circuit["selector_indices"], circuit["groups"],
                                                              circuit["num_gate_constraints"], cs["leaves"][row][: circuit["num_constants"]],
                                                              wires_c["leaves"][row], pih))
    unsalted = lambda c: [row[: len(c["polynomials"])] for row in c["leaves"]] if hiding else c["leaves"]  # noqa: E731  get_lde_values, oracle.rs:1007-1018
    quotient_polys = plonk_ref.compute_quotient_polys(unsalted(wires_c), cs["leaves"], unsalted(zs_c), circuit["num_constants"],
                                                      circuit["k_is"], betas, gammas, alphas, db, rate_bits, qdf, gate_terms)
    chunks = []
    for q in quotient_polys:
        assert all(c == 0 for c in q[n * qdf :]), "Quotient has failed, the vanishing polynomial is not divisible by Z_H"
        chunks += [q[k : k + n] for k in range(0, n * qdf, n)]
    quot_c = commit_from_coeffs(chunks, rate_bits, cap_height, salt_q)
    ch.observe_cap(quot_c["cap"])
    zeta = ch.get_extension_challenge()
    assert ext_pow(zeta, n) != (1, 0), "Opening point is in the subgroup."
    g_zeta = ext_mul((pyref.root_of_unity(db), 0), zeta)
    ev = lambda c, z: [plonk_ref.eval_ext2(p, z) for p in c["polynomials"]]  # noqa: E731
    cs_eval, zs_eval = ev(cs, zeta), ev(zs_c, zeta)
    openings = dict(constants=cs_eval[: circuit["num_constants"]], plonk_sigmas=cs_eval[circuit["num_constants"] :],
                    wires=ev(wires_c, zeta), plonk_zs=zs_eval[:nch], plonk_zs_next=ev(zs_c, g_zeta)[:nch],
                    partial_products=zs_eval[nch:], quotient_polys=ev(quot_c, zeta))
    for batch in fri_openings(openings):
        ch.observe_extension_elements(batch)
    opening_proof = fri_ref.prove_openings(fri_instance(circuit, zeta), [cs, wires_c, zs_c, quot_c], ch, fp)
    if trace is not None:
        trace.update(wires_commitment=wires_c, zs_partial_products=zs_pp, zs_partial_products_commitment=zs_c, betas=betas, gammas=gammas,
                     alphas=alphas, quotient_polys=quotient_polys, public_inputs_hash=pih)
    return dict(wires_cap=wires_c["cap"], plonk_zs_partial_products_cap=zs_c["cap"], quotient_polys_cap=quot_c["cap"],
                openings=openings, opening_proof=opening_proof, public_inputs=list(public_inputs))


# ---------------------------------------------------------------- verifier
def get_challenges(circuit, proof, pih):
    """plonk/get_challenges.rs:28-75"""
    nch = circuit["num_challenges"]
    ch = fri_ref.Challenger()
    ch.observe_elements(circuit["circuit_digest"])
    ch.observe_elements(pih)
    ch.observe_cap(proof["wires_cap"])
    betas, gammas = ch.get_n_challenges(nch), ch.get_n_challenges(nch)
    ch.observe_cap(proof["plonk_zs_partial_products_cap"])
    alphas = ch.get_n_challenges(nch)
    ch.observe_cap(proof["quotient_polys_cap"])
    zeta = ch.get_extension_challenge()
    for batch in fri_openings(proof["openings"]):
        ch.observe_extension_elements(batch)
    fri = fri_ref.fri_challenges(ch, proof["opening_proof"], circuit["degree_bits"], circuit["fri_params"])
    return dict(plonk_betas=betas, plonk_gammas=gammas, plonk_alphas=alphas, plonk_zeta=zeta, fri_challenges=fri)


def _scalar(x, k):
    return (x[0] * k % P, x[1] * k % P)


def gate_constraints_ext(circuit, local_constants, local_wires, pih):
    """evaluate_gate_constraints over the extension (plonk/vanishing_poly.rs:228-265) with
    Gate::eval_filtered (gates/gate.rs:86-107): written directly on F_{p^2} pairs, independently of
    plonk_ref's base-field gate closures."""
    num_selectors = len(circuit["groups"])
    consts = local_constants[num_selectors:]
    out = [(0, 0)] * circuit["num_gate_constraints"]
    for row, (kind, param) in enumerate(circuit["gates"]):
        si = circuit["selector_indices"][row]
        a, b = circuit["groups"][si]
        filt = (1, 0)
        for i in list(range(a, b)) + ([plonk_ref.UNUSED_SELECTOR] if num_selectors > 1 else []):
            if i != row:
                filt = ext_mul(filt, ext_sub((i, 0), local_constants[si]))
        if kind == "noop":
            cons = []
        elif kind == "constant":
            cons = [ext_sub(consts[i], local_wires[i]) for i in range(param)]
        elif kind == "public_input":
            cons = [ext_sub(local_wires[i], (pih[i], 0)) for i in range(4)]
        elif kind == "arithmetic":
            cons = []
            for i in range(param):
                m = ext_mul(ext_mul(local_wires[4 * i], local_wires[4 * i + 1]), consts[0])
                cons.append(ext_sub(local_wires[4 * i + 3], ext_add(m, ext_mul(local_wires[4 * i + 2], consts[1]))))
        else:
            cons = gates_ref.constraints(kind, param, consts, local_wires, pih, gates_ref.Ext)
        for k, c in enumerate(cons):
            out[k] = ext_add(out[k], ext_mul(filt, c))
    return out


def eval_vanishing_poly(circuit, x, openings, pih, betas, gammas, alphas):
    """plonk/vanishing_poly.rs:25-98"""
    n, qdf, num_routed = 1 << circuit["degree_bits"], circuit["quotient_degree_factor"], circuit["num_routed_wires"]
    num_prods = plonk_ref.num_partial_products(num_routed, qdf)
    constraint_terms = gate_constraints_ext(circuit, openings["constants"], openings["wires"], pih)
    # eval_l_0 (plonk_common.rs:57-67)
    l_0_x = (1, 0) if x == (1, 0) else ext_mul(ext_sub(ext_pow(x, n), (1, 0)), ext_inv(_scalar(ext_sub(x, (1, 0)), n)))
    z1, pp = [], []
    for i in range(circuit["num_challenges"]):
        z_x, z_gx = openings["plonk_zs"][i], openings["plonk_zs_next"][i]
        z1.append(ext_mul(l_0_x, ext_sub(z_x, (1, 0))))
        nums = [ext_add(ext_add(openings["wires"][j], _scalar(_scalar(x, circuit["k_is"][j]), betas[i])), (gammas[i], 0))
                for j in range(num_routed)]
        dens = [ext_add(ext_add(openings["wires"][j], _scalar(openings["plonk_sigmas"][j], betas[i])), (gammas[i], 0))
                for j in range(num_routed)]
        # check_partial_products (util/partial_products.rs:52-76)
        accs = [z_x] + openings["partial_products"][i * num_prods : (i + 1) * num_prods] + [z_gx]
        for c, k in enumerate(range(0, num_routed, qdf)):
            np_, dp = (1, 0), (1, 0)
            for v in nums[k : k + qdf]:
                np_ = ext_mul(np_, v)
            for v in dens[k : k + qdf]:
                dp = ext_mul(dp, v)
            pp.append(ext_sub(ext_mul(accs[c], np_), ext_mul(accs[c + 1], dp)))
    terms = z1 + pp + constraint_terms
    out = []
    for a in alphas:  # reduce_with_powers_multi (plonk_common.rs:97-114)
        acc = (0, 0)
        for t in reversed(terms):
            acc = ext_add(t, _scalar(acc, a))
        out.append(acc)
    return out


def verify(circuit, proof):
    """plonk/verifier.rs:15-120. Returns True or raises AssertionError."""
    pih = pyref.hash_no_pad(proof["public_inputs"])
    chal = get_challenges(circuit, proof, pih)
    op = proof["openings"]
    zeta = chal["plonk_zeta"]
    vanishing = eval_vanishing_poly(circuit, zeta, op, pih, chal["plonk_betas"], chal["plonk_gammas"], chal["plonk_alphas"])
    zeta_pow_deg = ext_pow(zeta, 1 << circuit["degree_bits"])
    z_h_zeta = ext_sub(zeta_pow_deg, (1, 0))
    qdf = circuit["quotient_degree_factor"]
    assert len(op["quotient_polys"]) == circuit["num_challenges"] * qdf
    for i in range(circuit["num_challenges"]):
        t = fri_ref.reduce_with_powers_ext(op["quotient_polys"][i * qdf : (i + 1) * qdf], zeta_pow_deg)
        assert vanishing[i] == ext_mul(z_h_zeta, t), "vanishing(zeta) != Z_H(zeta) * t(zeta)"
    caps = [circuit["constants_sigmas"]["cap"], proof["wires_cap"], proof["plonk_zs_partial_products_cap"], proof["quotient_polys_cap"]]
    return fri_ref.verify_fri_proof(fri_instance(circuit, zeta), fri_openings(op), chal["fri_challenges"], caps, proof["opening_proof"],
                                    circuit["degree_bits"], circuit["fri_params"])
