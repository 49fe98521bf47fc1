"""Synthetic circuits of a given shape with a satisfying witness, generated with numpy (bench and
large-size test support; not part of the product). Rows instantiate Noop / Constant{2} / PublicInput /
Arithmetic{num_routed/4}; copy constraints are random 2-cycles between arithmetic inputs.

gate_table="mini": the circuit's gate list is just those four, in one selector group.
gate_table="ed25519": the circuit DECLARES the whole 25-gate table of the plonky2-ed25519 circuit
(plonky2_gpu_amd/ed25519_circuit.py: 6 selector groups, 231 gate constraints) and instantiates the same four kinds
out of it. The quotient stage evaluates every gate of the list at every LDE point whichever gate a row uses (the
selector filters vanish on the subgroup only), so its cost is exactly that of the real circuit, and the witness is
still satisfying: the proof is valid. Selector columns follow selector_polynomials (plonk/circuit_builder.rs):
the row's gate index in its own group's column, UNUSED_SELECTOR = 2^32 - 1 in the others."""
import numpy as np

P = 0xFFFFFFFF00000001
_M32 = np.uint64(0xFFFFFFFF)
_EPS = np.uint64(0xFFFFFFFF)
_P = np.uint64(P)
_S32 = np.uint64(32)


def np_mul(a, b):
    """element-wise a*b mod p on uint64 arrays (canonical inputs and outputs)"""
    a, b = np.asarray(a, dtype=np.uint64), np.asarray(b, dtype=np.uint64)
    a0, a1, b0, b1 = a & _M32, a >> _S32, b & _M32, b >> _S32
    p00, p01, p10, p11 = a0 * b0, a0 * b1, a1 * b0, a1 * b1
    mid = p01 + (p00 >> _S32)
    mid2 = p10 + (mid & _M32)
    lo = (p00 & _M32) | ((mid2 & _M32) << _S32)
    hi = p11 + (mid >> _S32) + (mid2 >> _S32)
    hh, hl = hi >> _S32, hi & _M32
    t0 = lo - hh
    t0 = t0 - (lo < hh).astype(np.uint64) * _EPS
    t1 = hl * _EPS
    r = t0 + t1
    r = r + (r < t0).astype(np.uint64) * _EPS
    return np.where(r >= _P, r - _P, r)


def np_add(a, b):
    a, b = np.asarray(a, dtype=np.uint64), np.asarray(b, dtype=np.uint64)
    s = a + b
    s = s + (s < a).astype(np.uint64) * _EPS
    return np.where(s >= _P, s - _P, s)


def np_random(rng, shape):
    v = rng.integers(0, P, size=shape, dtype=np.uint64, endpoint=False)
    return v


def subgroup(degree_bits):
    w = pow(1753635133440165772, 1 << (32 - degree_bits), P)
    out = np.ones(1, dtype=np.uint64)
    for k in range(degree_bits):
        out = np.concatenate([out, np_mul(out, np.uint64(pow(w, 1 << k, P)))])
    return out


UNUSED_SELECTOR = (1 << 32) - 1


def make(degree_bits, num_wires=135, num_routed=80, num_constants=8, seed=1, num_copy_pairs=None, fri_params=None, gate_table="mini"):
    rng = np.random.default_rng(seed)
    n = 1 << degree_bits
    num_ops = num_routed // 4
    assert num_constants >= 3 and num_wires >= num_routed
    if gate_table == "ed25519":
        from plonky2_gpu_amd import ed25519_circuit as ed

        assert (num_wires, num_routed, num_constants) == (ed.NUM_WIRES, ed.NUM_ROUTED_WIRES, ed.NUM_CONSTANTS)
        table = dict(gates=list(ed.GATES), selector_indices=list(ed.SELECTOR_INDICES), groups=list(ed.GROUPS),
                     num_gate_constraints=ed.NUM_GATE_CONSTRAINTS)
        arith_row, first_const = 5, len(ed.GROUPS)  # ArithmeticGate{20} is entry 5; gate constants follow the 6 selectors
        assert ed.GATES[arith_row] == ("arithmetic", num_ops) and ed.GATES[1] == ("constant", 2)
    else:
        assert gate_table == "mini"
        table = dict(gates=[("noop", None), ("constant", 2), ("public_input", None), ("arithmetic", num_ops)],
                     selector_indices=[0, 0, 0, 0], groups=[(0, 4)], num_gate_constraints=max(num_ops, 4))
        arith_row, first_const = 3, 1
    public_inputs = [int(x) for x in np_random(rng, 3)]
    k_is = [pow(7, j, P) for j in range(num_routed)]
    row_gate = rng.choice(np.array([0, 1, 3, 3, 3, 3], dtype=np.int64), size=n)
    row_gate[0] = 2
    constants = np_random(rng, (num_constants, n))
    constants[0] = np.where(row_gate == 3, arith_row, row_gate).astype(np.uint64)  # the selector column of group 0
    constants[1:first_const] = np.uint64(UNUSED_SELECTOR)  # no row uses a gate of the other groups
    c0, c1 = constants[first_const], constants[first_const + 1]
    wires = np_random(rng, (num_wires, n))
    arith_rows = np.flatnonzero(row_gate == 3)
    # copy constraints: disjoint random pairs of arithmetic input cells
    in_cols = np.array([4 * i + k for i in range(num_ops) for k in range(3)], dtype=np.int64)
    total = arith_rows.size * in_cols.size
    npairs = min(num_copy_pairs if num_copy_pairs is not None else n // 4, total // 2)
    pick = rng.choice(total, size=2 * npairs, replace=False)
    rows, cols = arith_rows[pick // in_cols.size], in_cols[pick % in_cols.size]
    ra, ca, rb, cb = rows[:npairs], cols[:npairs], rows[npairs:], cols[npairs:]
    wires[cb, rb] = wires[ca, ra]
    sub = subgroup(degree_bits)
    sigmas = np.stack([np_mul(sub, np.uint64(k)) for k in k_is])
    k_arr = np.array(k_is, dtype=np.uint64)
    sigmas[ca, ra] = np_mul(k_arr[cb], sub[rb])
    sigmas[cb, rb] = np_mul(k_arr[ca], sub[ra])
    # gate outputs
    for i in range(num_ops):
        prod = np_mul(np_mul(wires[4 * i, arith_rows], wires[4 * i + 1, arith_rows]), c0[arith_rows])
        wires[4 * i + 3, arith_rows] = np_add(prod, np_mul(wires[4 * i + 2, arith_rows], c1[arith_rows]))
    const_rows = np.flatnonzero(row_gate == 1)
    wires[0, const_rows], wires[1, const_rows] = c0[const_rows], c1[const_rows]
    circuit = dict(degree_bits=degree_bits, num_wires=num_wires, num_routed_wires=num_routed, num_constants=num_constants,
                   num_challenges=2, quotient_degree_factor=8, k_is=k_is, constants=constants, sigmas=sigmas, **table,
                   fri_params=fri_params or dict(rate_bits=3, cap_height=4, reduction_arity_bits=constant_arity_bits(degree_bits, 3, 4),
                                                 proof_of_work_bits=16, num_query_rounds=28))
    return circuit, wires, public_inputs


def constant_arity_bits(degree_bits, rate_bits, cap_height, arity_bits=4, final_poly_bits=5):
    """FriReductionStrategy::ConstantArityBits(4, 5) (plonky2/src/fri/reduction_strategies.rs:38-48)"""
    out = []
    while degree_bits > final_poly_bits and degree_bits + rate_bits - arity_bits >= cap_height:
        out.append(arity_bits)
        degree_bits -= arity_bits
    return out


def set_public_input_row(wires, pih):
    """the PublicInputGate sits in row 0 (its wires must equal the public-inputs hash)"""
    for i in range(4):
        wires[i, 0] = pih[i]
