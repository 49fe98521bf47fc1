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
