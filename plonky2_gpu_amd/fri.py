"""Host-side mirror of PolynomialBatch::prove_openings (plonky2/src/fri/oracle.rs:1047-1112) and
fri_proof (plonky2/src/fri/prover.rs:24-260). Polynomials, codewords and trees stay in HBM; the host
only sees caps, challenges, the final polynomial and the queried leaves / Merkle paths."""
import ctypes

import numpy as np

from . import _lib
from .device import DeviceBuffer
from .merkle_tree import MerkleTree

P = 0xFFFFFFFF00000001
W = 7  # X^2 = 7, field/src/goldilocks_extensions.rs:19
COSET_SHIFT = 7


def ext_mul(x, y):
    return ((x[0] * y[0] + W * x[1] * y[1]) % P, (x[0] * y[1] + x[1] * y[0]) % P)


def ext_pow(x, e):
    acc = (1, 0)
    while e:
        if e & 1:
            acc = ext_mul(acc, x)
        x = ext_mul(x, x)
        e >>= 1
    return acc


def _pair(x):
    return np.array([int(x[0]) % P, int(x[1]) % P], dtype=np.uint64)


def _coset_lde_planar(ctx, d_coeffs, length, rate_bits, shift):
    """values of the extension polynomial on shift*H_{length << rate_bits}, bit-reversed, planar."""
    log_len = length.bit_length() - 1
    d_vals = DeviceBuffer(ctx, 2 * (length << rate_bits))
    _lib.call("gl_coset_lde_batch", d_coeffs.ptr, d_vals.ptr, 2, log_len, rate_bits, shift, length, length << rate_bits, ctx.ptr)
    return d_vals


def fri_committed_trees(ctx, d_coeffs, length, challenger, params):
    """fri/prover.rs:77-120. d_coeffs: planar extension coefficients (the non-zero 1/rate prefix of
    the reference's zero-padded vector). Returns (trees, final_coeffs)."""
    rate_bits, cap_height = params["rate_bits"], params["cap_height"]
    trees = []
    shift = COSET_SHIFT
    d_vals = _coset_lde_planar(ctx, d_coeffs, length, rate_bits, shift)
    arities = params["reduction_arity_bits"]
    for li, ab in enumerate(arities):
        arity = 1 << ab
        lde_len = length << rate_bits
        d_rows = DeviceBuffer(ctx, 2 * lde_len)
        _lib.call("gl_ext2_interleave", d_vals.ptr, lde_len, d_rows.ptr, ctx.ptr)
        n_leaves = lde_len >> ab
        d_dig = DeviceBuffer(ctx, max(8 * (n_leaves - (1 << cap_height)), 4))
        d_cap = DeviceBuffer(ctx, 4 << cap_height)
        _lib.call("gl_merkle_tree_from_leaves", d_rows.ptr, 2 * arity, n_leaves, cap_height, d_dig.ptr, d_cap.ptr, ctx.ptr)
        tree = MerkleTree(ctx, n_leaves, 2 * arity, cap_height, d_dig, d_cap, d_rows)
        challenger.observe_cap(tree.cap.tolist())
        trees.append(tree)
        beta = challenger.get_extension_challenge()
        d_new = DeviceBuffer(ctx, 2 * (length >> ab))
        _lib.call("gl_fri_fold", d_coeffs.ptr, length, ab, _pair(beta), d_new.ptr, ctx.ptr)
        length >>= ab
        d_coeffs = d_new
        shift = pow(shift, arity, P)
        d_vals.free()
        if li + 1 < len(arities):
            d_vals = _coset_lde_planar(ctx, d_coeffs, length, rate_bits, shift)
    planes = d_coeffs.download(0, 2 * length).reshape(2, length)
    final = [(int(a), int(b)) for a, b in zip(planes[0], planes[1])]
    challenger.observe_extension_elements(final)
    return trees, final


def fri_proof_of_work(ctx, challenger, params):
    """fri/prover.rs:122-171 — the grinding runs on the device and returns the smallest witness."""
    min_lz = params["proof_of_work_bits"] + (64 - P.bit_length())
    state = list(challenger.sponge_state)
    pos = len(challenger.input_buffer)
    for i, x in enumerate(challenger.input_buffer):
        state[i] = x
    st = np.array(state, dtype=np.uint64)
    w = ctypes.c_uint64()
    _lib.call("gl_fri_proof_of_work", st, pos, min_lz, ctypes.addressof(w), ctx.ptr)
    challenger.observe_element(w.value)
    resp = challenger.get_challenge()
    assert 64 - resp.bit_length() >= min_lz
    return w.value


def fri_prover_query_rounds(initial_trees, trees, challenger, n, params):
    """fri/prover.rs:173-260: leaves and Merkle paths fetched from HBM per query."""
    indices = [rand % n for rand in challenger.get_n_challenges(params["num_query_rounds"])]
    # one gather per tree for all queries instead of a leaf read and log(n) digest reads per query
    initial = [t.open_batch(indices) for t in initial_trees]
    steps, cur = [], list(indices)
    for i, t in enumerate(trees):
        cur = [x >> params["reduction_arity_bits"][i] for x in cur]
        steps.append(t.open_batch(cur))
    rounds = []
    for q in range(len(indices)):
        rounds.append(dict(
            initial_trees_proof=[(lv[q].tolist(), sib[q].tolist()) for lv, sib in initial],
            steps=[dict(evals=[(int(a), int(b)) for a, b in lv[q].reshape(-1, 2)], merkle_proof=sib[q].tolist()) for lv, sib in steps]))
    return rounds


def prove_openings(ctx, instance, oracles, challenger, params, timing=None):
    """PolynomialBatch::prove_openings(instance, oracles, challenger, fri_params) (fri/oracle.rs:1047-1112).
    instance["batches"] = [(point, [(oracle_index, polynomial_index), ...]), ...]; oracles are
    PolynomialBatch objects committed with leaf_major=True."""
    import time

    def stage(name, t0):
        if timing is not None:
            ctx.synchronize()
            timing[name] = timing.get(name, 0.0) + (time.perf_counter() - t0) * 1e3
        return time.perf_counter()

    t = time.perf_counter()
    alpha = challenger.get_extension_challenge()
    n = 1 << oracles[0].degree_log
    d_final = DeviceBuffer(ctx, 2 * n)
    first = True
    for point, polys in instance["batches"]:
        ptrs = np.array([oracles[oi].d_polynomials.ptr + pi * n * 8 for oi, pi in polys], dtype=np.uint64)
        d_ptrs = DeviceBuffer.from_host(ctx, ptrs)
        d_comp = DeviceBuffer(ctx, 2 * n)
        _lib.call("gl_fri_reduce_polys_base", d_ptrs.ptr, len(polys), n, _pair(alpha), d_comp.ptr, ctx.ptr)
        scale = ext_pow(alpha, len(polys))  # alpha.shift_poly (util/reducing.rs:103-106)
        _lib.call("gl_fri_divide_by_linear", d_comp.ptr, n, _pair(point), _pair(scale), 0 if first else 1,
                  d_final.ptr, ctx.ptr)
        ctx.synchronize()
        d_ptrs.free()
        d_comp.free()
        first = False
    n_lde = n << params["rate_bits"]
    t = stage("fri: combine + divide", t)
    trees, final_coeffs = fri_committed_trees(ctx, d_final, n, challenger, params)
    t = stage("fri: commit phase", t)
    pow_witness = fri_proof_of_work(ctx, challenger, params)
    t = stage("fri: proof of work", t)
    rounds = fri_prover_query_rounds([o.merkle_tree for o in oracles], trees, challenger, n_lde, params)
    stage("fri: query rounds", t)
    return dict(commit_phase_merkle_caps=[t.cap.tolist() for t in trees], query_round_proofs=rounds, final_poly=final_coeffs,
                pow_witness=pow_witness)
