"""Host-side mirror of plonky2/src/hash/merkle_tree.rs over device-resident leaves."""
import numpy as np

from . import _lib
from .device import DeviceBuffer


def _log2_strict(n):
    if n <= 0 or n & (n - 1):
        raise ValueError("not a power of two")
    return n.bit_length() - 1


class MerkleTree:
    """MerkleTree<F, PoseidonHash> (merkle_tree.rs:41-70): `digests` in the reference layout,
    `cap` = 2^cap_height subtree roots. Leaves stay in HBM."""

    def __init__(self, ctx, n_leaves, leaf_len, cap_height, digests_buf, cap_buf, leaves_buf=None, cols_buf=None, col_stride=None):
        self.ctx = ctx
        self.n_leaves = n_leaves
        self.leaf_len = leaf_len
        self.cap_height = cap_height
        self.d_digests = digests_buf
        self.d_cap = cap_buf
        self.d_leaves = leaves_buf  # leaf-major [n_leaves][leaf_len] or None
        self.d_cols = cols_buf  # column-major [leaf_len][col_stride] (the LDE the tree was built from) or None
        self.col_stride = col_stride
        self._digests = None
        self._cap = None

    @classmethod
    def new(cls, ctx, leaves, cap_height):
        """MerkleTree::new(leaves, cap_height) (merkle_tree.rs:283-319); leaves [n_leaves, leaf_len]."""
        lv = np.ascontiguousarray(leaves, dtype=np.uint64)
        n, ll = lv.shape
        if cap_height > _log2_strict(n):
            # the reference asserts (merkle_tree.rs:285-290)
            raise ValueError(f"cap_height={cap_height} should be at most log2(leaves.len())={_log2_strict(n)}")
        d_leaves = DeviceBuffer.from_host(ctx, lv)
        d_dig = DeviceBuffer(ctx, 4 * 2 * (n - (1 << cap_height)))
        d_cap = DeviceBuffer(ctx, 4 << cap_height)
        _lib.call("gl_merkle_tree_from_leaves", d_leaves.ptr, ll, n, cap_height, d_dig.ptr, d_cap.ptr, ctx.ptr)
        return cls(ctx, n, ll, cap_height, d_dig, d_cap, d_leaves)

    @property
    def digests(self):
        if self._digests is None:
            self._digests = self.d_digests.download(0, 4 * 2 * (self.n_leaves - (1 << self.cap_height))).reshape(-1, 4)
        return self._digests

    @property
    def cap(self):
        if self._cap is None:
            self._cap = self.d_cap.download(0, 4 << self.cap_height).reshape(-1, 4)
        return self._cap

    def get(self, i):
        """MerkleTree::get (merkle_tree.rs:385-391): leaf i, fetched from HBM."""
        if self.d_leaves is None:
            return self.open_batch([i])[0][0]
        return self.d_leaves.download(i * self.leaf_len, self.leaf_len)

    def prove(self, leaf_index):
        """MerkleTree::prove (merkle_tree.rs:392-440): sibling digests from the leaf up to the cap."""
        num_layers = _log2_strict(self.n_leaves) - self.cap_height
        assert leaf_index >> (self.cap_height + num_layers) == 0
        tree_len = 2 * (self.n_leaves - (1 << self.cap_height)) >> self.cap_height
        base = tree_len * (leaf_index >> num_layers)
        pair_index = leaf_index & ((1 << num_layers) - 1)
        siblings = []
        for i in range(num_layers):
            parity = pair_index & 1
            pair_index >>= 1
            siblings_index = (pair_index << (i + 1)) + (1 << i) - 1
            slot = base + 2 * siblings_index + (1 - parity)
            # one 32-byte read per layer straight from HBM unless the whole array is already on the host
            siblings.append(self._digests[slot] if self._digests is not None else self.d_digests.download(4 * slot, 4))
        return np.array(siblings, dtype=np.uint64).reshape(num_layers, 4)

    def open_batch(self, indices):
        """get(i) and prove(i) for many leaves with one launch and one copy (what
        fri_prover_query_round, fri/prover.rs:199-260, asks of every tree for every query).
        Returns (leaves [count, leaf_len], siblings [count, num_layers, 4])."""
        if self.d_leaves is not None:
            base, rs, es = self.d_leaves.ptr, self.leaf_len, 1
        elif self.d_cols is not None:
            base, rs, es = self.d_cols.ptr, 1, self.col_stride
        else:
            raise ValueError("the tree holds neither leaf-major rows nor the column-major matrix")
        idx = np.ascontiguousarray(indices, dtype=np.uint64)
        layers = _log2_strict(self.n_leaves) - self.cap_height
        leaves = np.empty((idx.size, self.leaf_len), dtype=np.uint64)
        sib = np.empty((idx.size, layers, 4), dtype=np.uint64)
        _lib.call("gl_merkle_open_batch", base, rs, es, self.leaf_len, self.n_leaves, self.cap_height,
                  self.d_digests.ptr if layers else None, idx, idx.size, leaves, sib if layers else None, self.ctx.ptr)
        return leaves, sib
