"""A small valid copy-constraint (permutation argument) instance for the quotient-path tests."""
import random

from oracle import pyref

P = pyref.P


def poly_eval(coeffs, x):
    acc = 0
    for c in reversed(coeffs):
        acc = (acc * x + c) % P
    return acc


def lde_leaves(columns, rate_bits):
    """PolynomialBatch::from_values' leaf-major LDE rows (fri/oracle.rs:709-731, 942-952) + coefficients."""
    coeffs = [pyref.fast_ntt(c, inverse=True) for c in columns]
    n_ext = len(columns[0]) << rate_bits
    lde = []
    for c in coeffs:
        scaled = [x * pow(pyref.GENERATOR, i, P) % P for i, x in enumerate(c)] + [0] * (n_ext - len(c))
        lde.append(pyref.fast_ntt(scaled))
    bits = n_ext.bit_length() - 1
    leaves = [[col[pyref.reverse_bits(i, bits)] for col in lde] for i in range(n_ext)]
    return coeffs, leaves


def make_instance(degree_bits=4, num_wires=12, num_routed=10, num_constants=2, num_challenges=2, seed=1, valid=True):
    rng = random.Random(seed)
    n = 1 << degree_bits
    w = pyref.root_of_unity(degree_bits)
    subgroup = [pow(w, i, P) for i in range(n)]
    k_is = [pow(pyref.GENERATOR, j, P) for j in range(num_routed)]  # get_unique_coset_shifts, field/src/cosets.rs:9-24
    # a random permutation of the routed cells, wires constant on its cycles
    cells = [(i, j) for j in range(num_routed) for i in range(n)]
    perm = cells[:]
    rng.shuffle(perm)
    sigma = dict(zip(cells, perm))
    wires = [[None] * n for _ in range(num_wires)]
    for start in cells:
        if wires[start[1]][start[0]] is None:
            v = rng.randrange(P)
            c = start
            while wires[c[1]][c[0]] is None:
                wires[c[1]][c[0]] = v
                c = sigma[c]
    for j in range(num_routed, num_wires):
        wires[j] = [rng.randrange(P) for _ in range(n)]
    if not valid:
        wires[0][0] = (wires[0][0] + 1) % P
    sigmas = [[k_is[sigma[(i, j)][1]] * subgroup[sigma[(i, j)][0]] % P for i in range(n)] for j in range(num_routed)]
    constants = [[rng.randrange(P) for _ in range(n)] for _ in range(num_constants)]
    betas = [rng.randrange(P) for _ in range(num_challenges)]
    gammas = [rng.randrange(P) for _ in range(num_challenges)]
    alphas = [rng.randrange(P) for _ in range(num_challenges)]
    return dict(n=n, degree_bits=degree_bits, subgroup=subgroup, k_is=k_is, wires=wires, sigmas=sigmas, constants=constants,
                betas=betas, gammas=gammas, alphas=alphas, num_routed=num_routed, num_constants=num_constants)


def make_circuit_instance(degree_bits=4, seed=1, two_groups=False, num_challenges=2, public_inputs=None):
    """A tiny real circuit: rows of NoopGate / ConstantGate{2} / PublicInputGate / ArithmeticGate{3}
    (plonky2/src/gates/{noop,constant,public_input,arithmetic_base}.rs) with a satisfying witness and
    copy constraints between arithmetic inputs. 12 wires, all routed."""
    from oracle import plonk_ref

    rng = random.Random(seed)
    n = 1 << degree_bits
    num_wires = num_routed = 12
    w = pyref.root_of_unity(degree_bits)
    subgroup = [pow(w, i, P) for i in range(n)]
    k_is = [pow(pyref.GENERATOR, j, P) for j in range(num_routed)]
    gates = [plonk_ref.noop_gate(), plonk_ref.constant_gate(2), plonk_ref.public_input_gate(), plonk_ref.arithmetic_gate(3)]
    if two_groups:
        groups, selector_indices = [(0, 2), (2, 4)], [0, 0, 1, 1]
    else:
        groups, selector_indices = [(0, 4)], [0, 0, 0, 0]
    num_selectors = len(groups)
    pih = [rng.randrange(P) for _ in range(4)] if public_inputs is None else pyref.hash_no_pad(public_inputs)
    row_gate = [rng.choice([0, 1, 3, 3, 3]) for _ in range(n)]
    row_gate[0] = 2  # the public-input gate sits in the first row
    # constants: selector columns, then the two gate constants
    sel = [[(row_gate[r] if selector_indices[row_gate[r]] == g else plonk_ref.UNUSED_SELECTOR) for r in range(n)]
           for g in range(num_selectors)]
    c0 = [rng.randrange(P) for _ in range(n)]
    c1 = [rng.randrange(P) for _ in range(n)]
    wires = [[rng.randrange(P) for _ in range(n)] for _ in range(num_wires)]
    # copy cycles among arithmetic inputs
    inputs = [(r, 4 * i + k) for r in range(n) if row_gate[r] == 3 for i in range(3) for k in range(3)]
    rng.shuffle(inputs)
    sigma = {(r, j): (r, j) for j in range(num_routed) for r in range(n)}
    pos = 0
    while pos + 1 < len(inputs):
        size = min(rng.choice([2, 2, 3]), len(inputs) - pos)
        cyc = inputs[pos : pos + size]
        pos += size
        v = rng.randrange(P)
        for t, cell in enumerate(cyc):
            wires[cell[1]][cell[0]] = v
            sigma[cell] = cyc[(t + 1) % size]
    for r in range(n):
        g = row_gate[r]
        if g == 3:
            for i in range(3):
                wires[4 * i + 3][r] = (wires[4 * i][r] * wires[4 * i + 1][r] % P * c0[r] + wires[4 * i + 2][r] * c1[r]) % P
        elif g == 1:
            wires[0][r], wires[1][r] = c0[r], c1[r]
        elif g == 2:
            for i in range(4):
                wires[i][r] = pih[i]
    sigmas = [[k_is[sigma[(i, j)][1]] * subgroup[sigma[(i, j)][0]] % P for i in range(n)] for j in range(num_routed)]
    return dict(n=n, degree_bits=degree_bits, subgroup=subgroup, k_is=k_is, wires=wires, sigmas=sigmas, constants=sel + [c0, c1],
                betas=[rng.randrange(P) for _ in range(num_challenges)], gammas=[rng.randrange(P) for _ in range(num_challenges)],
                alphas=[rng.randrange(P) for _ in range(num_challenges)], num_routed=num_routed, num_constants=num_selectors + 2,
                gates=gates, groups=groups, selector_indices=selector_indices, pih=pih, num_gate_constraints=4, row_gate=row_gate)


def make_circuit(degree_bits=4, seed=1, two_groups=False, arity_bits=(2, 1), rate_bits=3, cap_height=1, pow_bits=3, num_queries=3,
                 quotient_degree_factor=8, num_challenges=2):
    """The tiny circuit above as the dict oracle/prove_ref.py and plonky2_gpu_amd.prove() take
    (CommonCircuitData + ProverOnlyCircuitData of plonk/circuit_data.rs), plus its witness.
    quotient_degree_factor must be at least (filtered gate degree) - 1: 5 with one selector group
    (arithmetic degree 3 + filter degree 3), 4 with two groups (one filter factor + the unused-selector factor)."""
    from oracle import prove_ref

    rng = random.Random(seed * 7919)
    public_inputs = [rng.randrange(P) for _ in range(3)]
    inst = make_circuit_instance(degree_bits, seed, two_groups, public_inputs=public_inputs)
    cs = prove_ref.commit_from_values(inst["constants"] + inst["sigmas"], rate_bits, cap_height)
    circuit = dict(degree_bits=degree_bits, num_wires=12, num_routed_wires=12, num_constants=inst["num_constants"],
                   num_challenges=num_challenges, quotient_degree_factor=quotient_degree_factor, k_is=inst["k_is"],
                   gates=[("noop", None), ("constant", 2), ("public_input", None), ("arithmetic", 3)],
                   selector_indices=inst["selector_indices"], groups=inst["groups"], num_gate_constraints=4,
                   constants=inst["constants"], sigmas=inst["sigmas"],
                   fri_params=dict(rate_bits=rate_bits, cap_height=cap_height, reduction_arity_bits=list(arity_bits),
                                   proof_of_work_bits=pow_bits, num_query_rounds=num_queries),
                   constants_sigmas=cs, circuit_digest=prove_ref.circuit_digest(cs["cap"], degree_bits))
    return circuit, inst["wires"], public_inputs


FULL_GATES = [("noop", None), ("constant", 2), ("public_input", None), ("base_sum", (2, 8)), ("arithmetic", 3),
              ("base_sum", (4, 6)), ("comparison", (8, 4)), ("u32_add_many", (2, 2)), ("u32_arithmetic", 2),
              ("u32_subtraction", 2), ("u32_range_check", 2), ("random_access", (2, 2, 2)),
              ("poseidon", None)]
FULL_GROUPS = [(0, 5), (5, 9), (9, 12), (12, 13)]
FULL_SELECTOR_INDICES = [0] * 5 + [1] * 4 + [2] * 3 + [3]


# The gates a standard_recursion_config circuit (plonk/circuit_data.rs:60-90: 135 wires, 80 routed) is made of — the eight kinds of upstream
# plonky2 beyond the ed25519 list at the parameters that config gives them (new_from_config of each gate file), with the basic ones, sorted
# and grouped the way the builder does (by degree; degree(gate) + |group| <= quotient_degree_factor + 1 = 9, gates/selectors.rs:40-110):
# degrees 0 1 1 1 2 2 | 2 2 3 3 3 | 4 4 5 | 7.
RECURSION_GATES = [("noop", None), ("constant", 2), ("public_input", None), ("poseidon_mds", None), ("reducing", 43), ("reducing_extension", 32),
                   ("low_degree_interpolation", 4), ("base_sum", (2, 63)), ("arithmetic", 20), ("arithmetic_extension", 10), ("mul_extension", 13),
                   ("exponentiation", 66), ("high_degree_interpolation", 2), ("random_access", (4, 4, 2)), ("poseidon", None)]
RECURSION_GROUPS = [(0, 6), (6, 11), (11, 14), (14, 15)]
RECURSION_SELECTOR_INDICES = [0] * 6 + [1] * 5 + [2] * 3 + [3]


def make_recursion_circuit(degree_bits=4, seed=1, **kwargs):
    """A circuit of standard_recursion_config's shape whose rows use the gates such circuits are made of (RECURSION_GATES),
    each row honestly generated by the gate's own witness generator (oracle/gates_ref.fill_row)."""
    return make_full_circuit(degree_bits, seed, gates=RECURSION_GATES, groups=RECURSION_GROUPS, selector_indices=RECURSION_SELECTOR_INDICES, **kwargs)


def make_full_circuit(degree_bits=4, seed=1, arity_bits=(2, 1), rate_bits=3, cap_height=1, pow_bits=3, num_queries=2,
                      gates=None, groups=None, selector_indices=None):
    """Every gate kind of the ed25519 gate list (small parameters) in one circuit of
    the standard shape (135 wires, 80 routed): each row is one honestly generated gate row
    (oracle/gates_ref.fill_row), copy constraints tie random pairs of arithmetic inputs. The selector
    groups respect degree(gate) + |group| <= 9 like the reference's grouping (gates/selectors.rs).
    With gates / groups / selector_indices: the same construction over another gate list (make_recursion_circuit)."""
    from oracle import gates_ref, prove_ref

    FULL_GATES, FULL_GROUPS, FULL_SELECTOR_INDICES = gates or globals()["FULL_GATES"], groups or globals()["FULL_GROUPS"], selector_indices or globals()["FULL_SELECTOR_INDICES"]
    pub_row, arith_row = FULL_GATES.index(("public_input", None)), next(i for i, (k, _) in enumerate(FULL_GATES) if k == "arithmetic")
    arith_ops = FULL_GATES[arith_row][1]

    rng = random.Random(seed * 104729)
    n = 1 << degree_bits
    num_wires, num_routed = 135, 80
    num_selectors = len(FULL_GROUPS)
    public_inputs = [rng.randrange(P) for _ in range(3)]
    pih = pyref.hash_no_pad(public_inputs)
    row_gate = [rng.randrange(len(FULL_GATES)) for _ in range(n)]
    row_gate[0] = pub_row
    for k in range(len(FULL_GATES)):  # every gate at least once
        if k != pub_row:
            row_gate[1 + k % (n - 1)] = k
    row_gate = [g if (g != pub_row or r == 0) else 0 for r, g in enumerate(row_gate)]
    consts = [[rng.randrange(P) for _ in range(n)] for _ in range(2)]
    sel = [[(row_gate[r] if FULL_SELECTOR_INDICES[row_gate[r]] == g else 0xFFFFFFFF) for r in range(n)] for g in range(num_selectors)]
    wires = [[rng.randrange(P) for _ in range(n)] for _ in range(num_wires)]
    k_is = [pow(pyref.GENERATOR, j, P) for j in range(num_routed)]
    w = pyref.root_of_unity(degree_bits)
    subgroup = [pow(w, i, P) for i in range(n)]
    sigma = {(r, j): (r, j) for j in range(num_routed) for r in range(n)}
    # copy constraints between inputs of arithmetic rows, set before the rows are generated
    arith_rows = [r for r in range(n) if row_gate[r] == arith_row]
    inputs = [(r, 4 * i + k) for r in arith_rows for i in range(arith_ops) for k in range(3)]
    rng.shuffle(inputs)
    fixed = {}
    for a, b in zip(inputs[0::2], inputs[1::2]):
        v = rng.randrange(P)
        fixed[a] = fixed[b] = v
        sigma[a], sigma[b] = b, a
    for r in range(n):
        kind, param = FULL_GATES[row_gate[r]]
        row = gates_ref.fill_row(kind, param, rng, [consts[0][r], consts[1][r]], pih)
        if kind == "arithmetic":
            for i in range(arith_ops):
                for k in range(3):
                    if (r, 4 * i + k) in fixed:
                        row[4 * i + k] = fixed[(r, 4 * i + k)]
                row[4 * i + 3] = (row[4 * i] * row[4 * i + 1] % P * consts[0][r] + row[4 * i + 2] * consts[1][r]) % P
        for j, v in enumerate(row):
            wires[j][r] = v
    sigmas = [[k_is[sigma[(i, j)][1]] * subgroup[sigma[(i, j)][0]] % P for i in range(n)] for j in range(num_routed)]
    constants = sel + consts
    cs = prove_ref.commit_from_values(constants + sigmas, rate_bits, cap_height)
    ngc = max(gates_ref.num_constraints(k, p) for k, p in FULL_GATES)
    circuit = dict(degree_bits=degree_bits, num_wires=num_wires, num_routed_wires=num_routed, num_constants=num_selectors + 2,
                   num_challenges=2, quotient_degree_factor=8, k_is=k_is, gates=FULL_GATES, selector_indices=FULL_SELECTOR_INDICES,
                   groups=FULL_GROUPS, num_gate_constraints=ngc, constants=constants, sigmas=sigmas,
                   fri_params=dict(rate_bits=rate_bits, cap_height=cap_height, reduction_arity_bits=list(arity_bits),
                                   proof_of_work_bits=pow_bits, num_query_rounds=num_queries),
                   constants_sigmas=cs, circuit_digest=prove_ref.circuit_digest(cs["cap"], degree_bits))
    return circuit, wires, public_inputs
