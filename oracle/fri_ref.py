"""fri_ref.py — CPU restatement of the FRI opening pipeline (TEST INFRASTRUCTURE ONLY):
Challenger, PolynomialBatch::prove_openings, fri_proof (commit phase, proof of work, query rounds)
and the FRI verifier. Pure Python big ints, small sizes only; each function cites the reference.

Pin: the reference has no golden proofs, so the prover restatement is pinned by the
verifier restatement (verify_fri_proof accepts every proof produced here and rejects corrupted
ones, tests/test_oracle_fri.py), exactly how the reference tests itself (data.prove -> data.verify).
The only deviation from the reference is deliberate: proof-of-work grinding returns the SMALLEST
witness (the reference's rayon `find_any`, fri/prover.rs:151-163, is nondeterministic).
"""
from . import plonk_ref, pyref

P = pyref.P
SPONGE_RATE, SPONGE_WIDTH = 8, 12

ext_mul, ext_add = plonk_ref.ext2_mul, plonk_ref.ext2_add


def ext_sub(x, y):
    return ((x[0] - y[0]) % P, (x[1] - y[1]) % P)


def ext_pow(x, e):
    acc = (1, 0)
    while e:
        if e & 1:
            acc = ext_mul(acc, x)
        x = ext_mul(x, x)
        e >>= 1
    return acc


def ext_inv(x):
    """1/(a + bX) = (a - bX)/(a^2 - 7 b^2)"""
    a, b = x
    d = pow((a * a - plonk_ref.W * b * b) % P, P - 2, P)
    return (a * d % P, (-b) * d % P)


class Challenger:
    """plonky2/src/iop/challenger.rs:31-160 (duplex sponge, overwrite mode)."""

    def __init__(self):
        self.sponge_state = [0] * SPONGE_WIDTH
        self.input_buffer = []
        self.output_buffer = []

    def clone(self):
        c = Challenger()
        c.sponge_state, c.input_buffer, c.output_buffer = list(self.sponge_state), list(self.input_buffer), list(self.output_buffer)
        return c

    def permute(self, state):
        return pyref.poseidon(state)

    def observe_element(self, e):
        self.output_buffer = []
        self.input_buffer.append(e % P)
        if len(self.input_buffer) == SPONGE_RATE:
            self.duplexing()

    def observe_elements(self, es):
        for e in es:
            self.observe_element(e)

    def observe_extension_elements(self, es):
        for a, b in es:
            self.observe_element(a)
            self.observe_element(b)

    def observe_cap(self, cap):
        for h in cap:
            self.observe_elements(h)

    def get_challenge(self):
        if self.input_buffer or not self.output_buffer:
            self.duplexing()
        return self.output_buffer.pop()

    def get_n_challenges(self, n):
        return [self.get_challenge() for _ in range(n)]

    def get_extension_challenge(self):
        a, b = self.get_n_challenges(2)
        return (a, b)

    def duplexing(self):
        for i, x in enumerate(self.input_buffer):
            self.sponge_state[i] = x
        self.input_buffer = []
        self.sponge_state = self.permute(self.sponge_state)
        self.output_buffer = list(self.sponge_state[:SPONGE_RATE])


def reduce_with_powers_ext(terms, alpha):
    """plonk_common.rs:116-128 over the extension: sum terms[i] * alpha^i."""
    s = (0, 0)
    for t in reversed(terms):
        s = ext_add(ext_mul(s, alpha), t)
    return s


def reduce_polys_base(polys, alpha):
    """ReducingFactor::reduce_polys_base (util/reducing.rs:83-95): sum_j alpha^j * poly_j."""
    n = max(len(p) for p in polys)
    out = [(0, 0)] * n
    power = (1, 0)
    for p in polys:
        out = [ext_add(o, (c * power[0] % P, c * power[1] % P)) for o, c in zip(out, list(p) + [0] * (n - len(p)))]
        power = ext_mul(power, alpha)
    return out


def divide_by_linear(coeffs, z):
    """PolynomialCoeffs::divide_by_linear (field/src/polynomial/division.rs:75-88)."""
    bs, acc = [], (0, 0)
    for c in reversed(coeffs):
        acc = ext_add(ext_mul(acc, z), c)
        bs.append(acc)
    bs.pop()
    bs.reverse()
    return bs


def ext_coset_fft(coeffs, shift):
    """coset_fft over the extension = component-wise base transform (twiddles are base elements)."""
    sc, r = [], 1
    for a, b in coeffs:
        sc.append((a * r % P, b * r % P))
        r = r * shift % P
    fa = pyref.fast_ntt([c[0] for c in sc])
    fb = pyref.fast_ntt([c[1] for c in sc])
    return list(zip(fa, fb))


def flatten(ext_values):
    return [x for v in ext_values for x in v]


def reverse_index_bits(v):
    bits = len(v).bit_length() - 1
    return [v[pyref.reverse_bits(i, bits)] for i in range(len(v))]


def fri_committed_trees(coeffs, values, challenger, params):
    """fri/prover.rs:77-120"""
    trees = []
    shift = pyref.GENERATOR
    for arity_bits in params["reduction_arity_bits"]:
        arity = 1 << arity_bits
        values = reverse_index_bits(values)
        leaves = [flatten(values[k : k + arity]) for k in range(0, len(values), arity)]
        digests, cap = pyref.merkle_tree(leaves, params["cap_height"])
        challenger.observe_cap(cap)
        trees.append(dict(leaves=leaves, digests=digests, cap=cap))
        beta = challenger.get_extension_challenge()
        coeffs = [reduce_with_powers_ext(coeffs[k : k + arity], beta) for k in range(0, len(coeffs), arity)]
        shift = pow(shift, arity, P)
        values = ext_coset_fft(coeffs, shift)
    coeffs = coeffs[: len(coeffs) >> params["rate_bits"]]
    challenger.observe_extension_elements(coeffs)
    return trees, coeffs


def fri_proof_of_work(challenger, params):
    """fri/prover.rs:122-171, smallest witness (deterministic)."""
    min_lz = params["proof_of_work_bits"] + (64 - P.bit_length())
    state = list(challenger.sponge_state)
    pos = len(challenger.input_buffer)
    for i, x in enumerate(challenger.input_buffer):
        state[i] = x
    cand = 0
    while True:
        s = list(state)
        s[pos] = cand
        resp = pyref.poseidon(s)[SPONGE_RATE - 1]
        if 64 - resp.bit_length() >= min_lz:
            break
        cand += 1
    challenger.observe_element(cand)
    resp = challenger.get_challenge()
    assert 64 - resp.bit_length() >= min_lz
    return cand


def merkle_prove(digests, n_leaves, cap_height, leaf_index):
    """MerkleTree::prove (hash/merkle_tree.rs:392-440)"""
    num_layers = (n_leaves.bit_length() - 1) - cap_height
    tree_len = len(digests) >> cap_height
    tree = digests[tree_len * (leaf_index >> num_layers) : tree_len * ((leaf_index >> num_layers) + 1)]
    pair_index = leaf_index & ((1 << num_layers) - 1)
    sib = []
    for i in range(num_layers):
        parity = pair_index & 1
        pair_index >>= 1
        sib.append(tree[2 * ((pair_index << (i + 1)) + (1 << i) - 1) + (1 - parity)])
    return sib


def merkle_verify(leaf, index, cap, siblings):
    """verify_merkle_proof_to_cap (hash/merkle_proofs.rs:53-80)"""
    cur = pyref.hash_or_noop(leaf)
    for s in siblings:
        cur = pyref.two_to_one(s, cur) if index & 1 else pyref.two_to_one(cur, s)
        index >>= 1
    return cur == list(cap[index])


def fri_prover_query_rounds(initial_trees, trees, challenger, n, params):
    """fri/prover.rs:173-260"""
    rounds = []
    for rand in challenger.get_n_challenges(params["num_query_rounds"]):
        x_index = rand % n
        initial = [(list(t["leaves"][x_index]), merkle_prove(t["digests"], len(t["leaves"]), params["cap_height"], x_index))
                   for t in initial_trees]
        steps = []
        for i, t in enumerate(trees):
            ab = params["reduction_arity_bits"][i]
            leaf = t["leaves"][x_index >> ab]
            evals = [(leaf[2 * k], leaf[2 * k + 1]) for k in range(len(leaf) // 2)]
            steps.append(dict(evals=evals, merkle_proof=merkle_prove(t["digests"], len(t["leaves"]), params["cap_height"], x_index >> ab)))
            x_index >>= ab
        rounds.append(dict(initial_trees_proof=initial, steps=steps))
    return rounds


def prove_openings(instance, oracles, challenger, params):
    """PolynomialBatch::prove_openings (fri/oracle.rs:1047-1112) + fri_proof (fri/prover.rs:24-70).
    instance = {"batches": [(point_ext, [(oracle_index, polynomial_index), ...]), ...]};
    oracles[i] = {"polynomials": [coeff lists], "leaves", "digests", "cap"} (from a commit)."""
    alpha = challenger.get_extension_challenge()
    final_poly = []
    for point, polys in instance["batches"]:
        comp = reduce_polys_base([oracles[oi]["polynomials"][pi] for oi, pi in polys], alpha)
        quotient = divide_by_linear(comp, point)
        scale = ext_pow(alpha, len(polys))  # alpha.shift_poly: count = polynomials reduced in this batch
        final_poly = [ext_mul(c, scale) for c in final_poly]
        final_poly = [ext_add(a, b) for a, b in zip(final_poly + [(0, 0)] * (len(quotient) - len(final_poly)), quotient)]
    final_poly = [(0, 0)] + final_poly  # multiply by X (oracle.rs:1085-1087)
    n_lde = len(final_poly) << params["rate_bits"]
    lde_coeffs = final_poly + [(0, 0)] * (n_lde - len(final_poly))
    lde_values = ext_coset_fft(lde_coeffs, pyref.GENERATOR)
    trees, final_coeffs = fri_committed_trees(lde_coeffs, lde_values, challenger, params)
    pow_witness = fri_proof_of_work(challenger, params)
    rounds = fri_prover_query_rounds(oracles, trees, challenger, n_lde, params)
    return dict(commit_phase_merkle_caps=[t["cap"] for t in trees], query_round_proofs=rounds, final_poly=final_coeffs,
                pow_witness=pow_witness)


# ---------------------------------------------------------------- verifier
def fri_challenges(challenger, proof, degree_bits, params):
    """Challenger::fri_challenges (fri/challenges.rs:24-66)"""
    lde_size = 1 << (degree_bits + params["rate_bits"])
    alpha = challenger.get_extension_challenge()
    betas = []
    for cap in proof["commit_phase_merkle_caps"]:
        challenger.observe_cap(cap)
        betas.append(challenger.get_extension_challenge())
    challenger.observe_extension_elements(proof["final_poly"])
    challenger.observe_element(proof["pow_witness"])
    pow_response = challenger.get_challenge()
    indices = [challenger.get_challenge() % lde_size for _ in range(params["num_query_rounds"])]
    return dict(fri_alpha=alpha, fri_betas=betas, fri_pow_response=pow_response, fri_query_indices=indices)


def compute_evaluation(x, x_index_within_coset, arity_bits, evals, beta):
    """fri/verifier.rs:21-47: interpolate {(x g^i, P(x g^i))} and evaluate at beta (Lagrange)."""
    arity = 1 << arity_bits
    g = pyref.root_of_unity(arity_bits)
    evals = reverse_index_bits(list(evals))
    rev = pyref.reverse_bits(x_index_within_coset, arity_bits)
    coset_start = x * pow(g, arity - rev, P) % P
    pts = [((coset_start * pow(g, i, P)) % P, 0) for i in range(arity)]
    total = (0, 0)
    for i, (xi, yi) in enumerate(zip(pts, evals)):
        num, den = (1, 0), (1, 0)
        for j, xj in enumerate(pts):
            if j != i:
                num = ext_mul(num, ext_sub(beta, xj))
                den = ext_mul(den, ext_sub(xi, xj))
        total = ext_add(total, ext_mul(yi, ext_mul(num, ext_inv(den))))
    return total


def verify_fri_proof(instance, openings, challenges, initial_caps, proof, degree_bits, params):
    """verify_fri_proof (fri/verifier.rs:63-113) with fri_verifier_query_round (:177-258) and
    fri_combine_initial (:127-175). openings[b] = ext values opened at batch b's point, in batch
    order. Returns True / raises AssertionError."""
    n = 1 << (degree_bits + params["rate_bits"])
    log_n = degree_bits + params["rate_bits"]
    min_lz = params["proof_of_work_bits"] + (64 - P.bit_length())
    assert 64 - challenges["fri_pow_response"].bit_length() >= min_lz, "Invalid proof of work witness."
    assert len(proof["query_round_proofs"]) == params["num_query_rounds"]
    alpha = challenges["fri_alpha"]
    reduced_openings = [reduce_with_powers_ext(vals, alpha) for vals in openings]  # PrecomputedReducedOpenings :268-280
    for x_index, rp in zip(challenges["fri_query_indices"], proof["query_round_proofs"]):
        for (evals, mp), cap in zip(rp["initial_trees_proof"], initial_caps):
            assert merkle_verify(evals, x_index, cap, mp), "initial Merkle proof"
        subgroup_x = pyref.GENERATOR * pow(pyref.root_of_unity(log_n), pyref.reverse_bits(x_index, log_n), P) % P
        # fri_combine_initial
        s = (0, 0)
        for (point, polys), red_open in zip(instance["batches"], reduced_openings):
            evals = [(rp["initial_trees_proof"][oi][0][pi], 0) for oi, pi in polys]
            reduced = reduce_with_powers_ext(evals, alpha)
            numerator = ext_sub(reduced, red_open)
            denominator = ext_sub((subgroup_x, 0), point)
            s = ext_mul(s, ext_pow(alpha, len(polys)))
            s = ext_add(s, ext_mul(numerator, ext_inv(denominator)))
        old_eval = ext_mul(s, (subgroup_x, 0))
        xi = x_index
        for i, ab in enumerate(params["reduction_arity_bits"]):
            arity = 1 << ab
            evals = rp["steps"][i]["evals"]
            coset_index, within = xi >> ab, xi & (arity - 1)
            assert tuple(evals[within]) == tuple(old_eval), "FRI consistency"
            old_eval = compute_evaluation(subgroup_x, within, ab, evals, challenges["fri_betas"][i])
            assert merkle_verify(flatten(evals), coset_index, proof["commit_phase_merkle_caps"][i], rp["steps"][i]["merkle_proof"])
            subgroup_x = pow(subgroup_x, arity, P)
            xi = coset_index
        acc = (0, 0)
        for c in reversed(proof["final_poly"]):
            acc = ext_add(ext_mul(acc, (subgroup_x, 0)), c)
        assert tuple(acc) == tuple(old_eval), "Final polynomial evaluation is invalid."
    return True
