"""One rank of a column-sharded commit (plonky2_gpu_amd.dist.sharded_commit_from_values), launched by
tests/test_gpu_dist.py as  python -m torch.distributed.run --nproc-per-node W tests/dist_sharded_commit.py .
All ranks share device 0 (the GPU box has one GPU) and the exchange is staged through host memory over gloo; with
one GPU per rank and PLONKY2_DIST_BACKEND=nccl the same code exchanges device buffers over RCCL.
Every rank checks its part against the oracle's commit of the WHOLE matrix: its columns' coefficients and LDE, its
leaf range of all columns, its contiguous block of the digest buffer, and the gathered cap."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import plonky2_gpu_amd as pg  # noqa: E402
from oracle import oracle as o  # noqa: E402
from plonky2_gpu_amd.dist import ProverGroup, shard_range, sharded_commit_from_values, sharded_open_batch  # noqa: E402


def check_against_single_device_commit(g, ctx, sc, vals, total_cols, log_n, rate_bits, cap_height):
    """Sizes at which the oracle's leaf matrix would not fit comfortably in host memory per rank: the same trace
    committed whole on this rank's device (itself checked against the oracle at these sizes by tests/test_gpu_merkle.py)."""
    import time

    n_ext = 1 << (log_n + rate_bits)
    whole = pg.PolynomialBatch.from_values(ctx, vals, rate_bits, False, cap_height, leaf_major=False)
    assert (sc.cap == whole.merkle_tree.cap).all(), "cap"
    L = sc.leaves_per_rank
    for c in (0, total_cols // 2, total_cols - 1):
        got = sc.d_leaves.download(c * L, L)
        assert (got == whole.d_lde.download(c * n_ext + sc.leaf_lo, L)).all(), ("leaf range, column", c)
    rng = np.random.default_rng(5 + g.rank)
    for slot in [0, sc.num_digests - 1] + [int(x) for x in rng.integers(0, sc.num_digests, size=100)]:
        assert (sc.d_digests.download(4 * slot, 4) == whole.merkle_tree.d_digests.download(4 * (sc.digest_lo + slot), 4)).all(), slot
    g.barrier()
    print(f"rank {g.rank}/{g.world}: {total_cols} x 2^{log_n} sharded commit equals the single-device commit (cap, leaf range, sampled digests) ok")
    ctx.close()
    g.close()


def main():
    total_cols, log_n, rate_bits, cap_height = (int(x) for x in sys.argv[1:5]) if len(sys.argv) > 4 else (24, 10, 3, 4)
    g = ProverGroup(backend=os.environ.get("PLONKY2_DIST_BACKEND", "gloo"))
    ndev = pg.load().gl_device_count()
    ctx = pg.Context(g.local_rank % ndev)
    n, n_ext = 1 << log_n, 1 << (log_n + rate_bits)
    vals = o.random_field((total_cols, n), seed=31337)  # the same trace on every rank; each keeps only its columns
    lo, hi = shard_range(total_cols, g.world, g.rank)
    d_vals = pg.DeviceBuffer.from_host(ctx, np.ascontiguousarray(vals[lo:hi]))
    sc = sharded_commit_from_values(g, ctx, d_vals, lo, hi, total_cols, log_n, rate_bits, cap_height)

    # openings: every rank asks for the same global leaves (first, last, one per rank's range, random ones) and gets
    # leaf + path from whichever rank owns it; each must verify against the gathered cap with the oracle's verifier
    rng = np.random.default_rng(99)
    idx = [0, n_ext - 1] + [q * (n_ext // g.world) + 3 % (n_ext // g.world) for q in range(g.world)] + [int(x) for x in rng.integers(0, n_ext, size=20)]
    op_leaves, op_sib = sharded_open_batch(g, ctx, sc, idx)
    for k, i in enumerate(idx):
        assert o.merkle_verify(op_leaves[k], i, sc.cap, op_sib[k]), ("opening", i)
    bad = op_leaves[0].copy()
    bad[0] ^= np.uint64(1)
    assert not o.merkle_verify(bad, idx[0], sc.cap, op_sib[0])
    if total_cols * n_ext > (1 << 26):
        return check_against_single_device_commit(g, ctx, sc, vals, total_cols, log_n, rate_bits, cap_height)
    exp = o.commit_from_values(vals, rate_bits, cap_height, threads=2)
    coeffs, leaves, digests, cap = (o.canon(exp[k]) for k in ("coeffs", "leaves", "digests", "cap"))
    assert (sc.cap == cap).all(), "cap"
    assert (op_leaves == leaves[idx]).all(), "opened leaves"
    assert (sc.d_coeffs.download(0, (hi - lo) * n).reshape(hi - lo, n) == coeffs[lo:hi]).all(), "coefficients"
    assert (sc.d_lde.download(0, (hi - lo) * n_ext).reshape(hi - lo, n_ext) == leaves.T[lo:hi]).all(), "LDE columns"
    L = sc.leaves_per_rank
    got = sc.d_leaves.download(0, total_cols * L).reshape(total_cols, L)
    assert (got == leaves[sc.leaf_lo:sc.leaf_lo + L].T).all(), "leaf range of all columns"
    if sc.num_digests:
        d = sc.d_digests.download(0, 4 * sc.num_digests).reshape(-1, 4)
        assert (d == digests[sc.digest_lo:sc.digest_lo + sc.num_digests]).all(), "digest block"
    assert sum(g.gather_caps(np.array([[sc.num_digests, 0, 0, 0]], dtype=np.uint64))[q][0, 0] for q in range(g.world)) == digests.shape[0]
    g.barrier()
    print(f"rank {g.rank}/{g.world}: columns [{lo},{hi}), leaves [{sc.leaf_lo},{sc.leaf_lo + L}), digests [{sc.digest_lo},+{sc.num_digests}) ok")
    ctx.close()
    g.close()


if __name__ == "__main__":
    main()
