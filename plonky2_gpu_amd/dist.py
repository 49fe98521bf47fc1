"""One process per GPU: work partitioning and the only exchange the path has.

The hot path shards by independent units (columns of a batch transform, whole commitments /
proofs in the batch-of-proofs case): no data-path collective exists. What the ranks
do exchange is (a) a barrier and a max-reduce of elapsed time for measurement and (b) the final
2^cap_height x 32 B Merkle caps gathered to rank 0. On a GPU node the process group is created
with backend "nccl" (RCCL over xGMI); on CPU (tests) with "gloo". Rendezvous uses 127.0.0.1.
"""
import os

import numpy as np


def shard_range(n_units, world, rank):
    """Contiguous, balanced [lo, hi) slice of n_units for `rank` (first n_units % world ranks get one more)."""
    base, rem = divmod(n_units, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class ProverGroup:
    def __init__(self, backend=None, device_index=None, force=False, dry_device_path=False):
        """`force` creates the process group even for one rank (how the RCCL route is exercised on a one-GPU box).
        `dry_device_path` (with a host backend such as gloo, ranks sharing a device): the exchange builds its send and receive
        tensors over the library's own device pointers exactly as the RCCL route does (device_tensor) and stages them through
        host memory only for the transport itself — everything but the backend string is the code a node with a device per rank runs."""
        self.dry_device_path = bool(dry_device_path)
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.td = None
        self.backend = None
        if self.world > 1 or force:
            import torch
            import torch.distributed as td

            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
            self.backend = backend or "gloo"
            self.device_index = self.local_rank if device_index is None else int(device_index)
            if self.backend == "nccl" or self.dry_device_path:
                # torch brings its own copy of the HIP runtime; it must come up BEFORE the first HIP call of libplonky2_hip in this
                # process (the other order fails with "No HIP GPUs are available"): form the group first, create contexts after
                try:
                    torch.cuda.set_device(self.device_index)
                except RuntimeError as e:
                    raise RuntimeError("torch could not initialise the GPU; if libplonky2_hip was used first in this process, create the "
                                       "ProverGroup before the first library call (plonky2_gpu_amd.dist): %s" % e) from e
            # gloo announces its connections on the C++ stdout; rank 0's stdout carries the one JSON line of bench.py, so
            # the file descriptor (not just sys.stdout) points at stderr while the group is being formed
            import sys

            sys.stdout.flush()
            saved = os.dup(1)
            try:
                os.dup2(2, 1)
                kw = {}
                if self.backend == "nccl":  # bind the communicator to this rank's GPU instead of letting torch guess it from the rank
                    kw["device_id"] = torch.device("cuda", self.device_index)
                td.init_process_group(backend=self.backend, rank=self.rank, world_size=self.world, **kw)
                td.barrier()
            finally:
                os.dup2(saved, 1)
                os.close(saved)
            self.td, self.torch = td, torch

    def _dev(self):
        return self.torch.device("cuda", self.device_index) if self.backend == "nccl" else self.torch.device("cpu")

    def barrier(self):
        if self.td:
            self.td.barrier()

    def max(self, x):
        if not self.td:
            return float(x)
        t = self.torch.tensor([float(x)], dtype=self.torch.float64, device=self._dev())
        self.td.all_reduce(t, op=self.td.ReduceOp.MAX)
        return float(t[0])

    def sum(self, x):
        if not self.td:
            return float(x)
        t = self.torch.tensor([float(x)], dtype=self.torch.float64, device=self._dev())
        self.td.all_reduce(t, op=self.td.ReduceOp.SUM)
        return float(t[0])

    def gather_caps(self, cap):
        """all_gather of each rank's Merkle cap ([2^h, 4] u64). Returns the list ordered by rank."""
        cap = np.ascontiguousarray(cap, dtype=np.uint64)
        if not self.td:
            return [cap]
        mine = self.torch.from_numpy(cap.view(np.int64).copy()).to(self._dev())
        outs = [self.torch.empty_like(mine) for _ in range(self.world)]
        self.td.all_gather(outs, mine)
        return [o.cpu().numpy().view(np.uint64).reshape(cap.shape) for o in outs]

    def close(self):
        if self.td:
            self.td.destroy_process_group()
            self.td = None


class _DevicePointer:
    """A raw device pointer as an object torch.as_tensor understands (CUDA array interface v2): lets torch.distributed
    send from and receive into the library's own buffers without a copy."""

    def __init__(self, ptr, n_elems):
        self.__cuda_array_interface__ = {"shape": (int(n_elems),), "typestr": "<i8", "data": (int(ptr), False), "version": 2, "strides": None}


def device_tensor(torch, ptr, n_elems, device_index):
    return torch.as_tensor(_DevicePointer(ptr, n_elems), device=torch.device("cuda", device_index))


class ShardedCommit:
    """One rank's part of a commitment whose columns are spread over the ranks (second row):
    `d_coeffs` / `d_lde` hold this rank's columns [my_cols][n] / [my_cols][n_ext]; `d_leaves` is the column-major
    block [total_cols][leaves_per_rank] of ALL columns for this rank's leaf range [leaf_lo, leaf_lo + leaves_per_rank);
    `d_digests` is this rank's contiguous block of the tree's digest buffer (the reference lays the buffer out per
    cap subtree, hash/merkle_tree.rs:210-244, and the subtrees of a rank are adjacent), starting at digest
    `digest_lo`; `cap` is the whole tree's cap, identical on every rank."""

    def __init__(self, **kw):
        self.__dict__.update(kw)


EXCHANGE_CHUNK_COLS = 16  # columns per LDE + pack + send step: the LDE of chunk k+1 runs under the exchange of chunk k


class ShardedCommitPlan:
    """Everything a column-sharded commit of one shape needs, allocated ONCE: the LDE of this rank's columns, the leaf
    block of all columns for this rank's leaf range, the pack buffer, the digest block and the cap (round 2 allocated and
    freed four device buffers per call). `commit(d_values)` may be called any number of times (a prover commits several
    batches of the same shape); the buffers belong to the plan and are overwritten by the next call."""

    def __init__(self, group, ctx, total_cols, log_n, rate_bits, cap_height):
        from .device import DeviceBuffer

        W, r = group.world, group.rank
        if W & (W - 1) or W > (1 << cap_height):
            raise ValueError("the number of ranks must be a power of two and at most 2^cap_height (whole cap subtrees per rank)")
        self.group, self.ctx = group, ctx
        self.total_cols, self.log_n, self.rate_bits, self.cap_height = total_cols, log_n, rate_bits, cap_height
        self.n, self.n_ext = 1 << log_n, 1 << (log_n + rate_bits)
        self.bounds = [shard_range(total_cols, W, q) for q in range(W)]
        self.col_lo, self.col_hi = self.bounds[r]
        self.mine = self.col_hi - self.col_lo
        self.L = self.n_ext // W  # leaves per rank
        self.local_cap_height = cap_height - (W.bit_length() - 1)
        self.n_dig = 2 * (self.L - (1 << self.local_cap_height))
        self.d_lde = DeviceBuffer(ctx, max(self.mine, 1) * self.n_ext)
        self.d_leaves = DeviceBuffer(ctx, total_cols * self.L)
        self.d_packed = DeviceBuffer(ctx, W * max(self.mine, 1) * self.L) if W > 1 else None
        self.d_digests = DeviceBuffer(ctx, 4 * max(self.n_dig, 1))
        self.d_cap = DeviceBuffer(ctx, 4 << self.local_cap_height)
        self._host = {}  # gloo: staging tensors per (peer, chunk), reused

    def free(self):
        for b in (self.d_lde, self.d_leaves, self.d_packed, self.d_digests, self.d_cap):
            if b is not None:
                b.free()

    @staticmethod
    def chunks(cols):
        return [(c0, min(c0 + EXCHANGE_CHUNK_COLS, cols)) for c0 in range(0, cols, EXCHANGE_CHUNK_COLS)]

    def commit(self, d_values):
        from . import _lib

        g, ctx = self.group, self.ctx
        W, r, L, mine, n, n_ext = g.world, g.rank, self.L, self.mine, self.n, self.n_ext
        _lib.call("gl_ntt_batch", d_values.ptr, mine, self.log_n, n, 1, 0, ctx.ptr)
        if W == 1:
            _lib.call("gl_coset_lde_batch", d_values.ptr, self.d_lde.ptr, mine, self.log_n, self.rate_bits, 7, n, n_ext, ctx.ptr)
            _lib.call("gl_memcpy_d2d", self.d_leaves.ptr, self.d_lde.ptr, 8 * mine * n_ext, ctx.ptr)
        else:
            td, torch = g.td, g.torch
            on_device = g.backend == "nccl"
            dry = g.dry_device_path and not on_device  # device tensors over the library's pointers, transported through host memory
            reqs, staged = [], []
            # The path's ONE exchange, chunk by chunk: LDE of sixteen of my columns -> one pack launch groups, for every rank,
            # the leaf range it hashes of those columns (slice for rank q contiguous at packed[q][c0:c1][L]) -> the sends of
            # that chunk are posted (point to point, every link of an xGMI mesh carries one pair) and the next chunk's LDE is
            # queued on the library's stream while they travel: the exchange runs under the remaining LDE.
            my_chunks = self.chunks(mine)
            peer_chunks = {q: self.chunks(self.bounds[q][1] - self.bounds[q][0]) for q in range(W) if q != r}
            steps = max([len(my_chunks)] + [len(v) for v in peer_chunks.values()])
            for step in range(steps):
                # step `step`: my chunk goes out, every peer's chunk of the same number comes in — one group per step on
                # every rank, in the same order everywhere (sends and their receives in one group: no rank waits for a
                # receive that its peer has queued behind a send)
                ops = []
                if step < len(my_chunks):
                    c0, c1 = my_chunks[step]
                    k = c1 - c0
                    _lib.call("gl_coset_lde_batch", d_values.at(c0 * n), self.d_lde.at(c0 * n_ext), k, self.log_n, self.rate_bits, 7, n, n_ext, ctx.ptr)
                    # pack buffer layout [chunk][q][k][L]: chunk c0 starts at W * c0 * L, its slice for rank q is contiguous
                    base = W * c0 * L
                    _lib.call("gl_pack_leaf_ranges", self.d_lde.at(c0 * n_ext), n_ext, k, L, W, self.d_packed.at(base), ctx.ptr)
                    # my own columns of my own leaf range stay on the device
                    _lib.call("gl_memcpy_d2d", self.d_leaves.at((self.col_lo + c0) * L), self.d_packed.at(base + r * k * L), 8 * k * L, ctx.ptr)
                    ctx.synchronize()  # the sends read what the pack wrote (RCCL runs on its own stream)
                    for q in range(W):
                        if q == r:
                            continue
                        if on_device:  # zero copy: RCCL sends from the pack buffer
                            send = device_tensor(torch, self.d_packed.at(base + q * k * L), k * L, g.device_index)
                        elif dry:  # the same tensor over the pack buffer, copied to the host by torch for the host transport
                            send = device_tensor(torch, self.d_packed.at(base + q * k * L), k * L, g.device_index).cpu()
                        else:  # gloo: staged through host memory
                            send = self._host_tensor(("s", q, c0), k * L)
                            _lib.call("gl_memcpy_d2h", send.data_ptr(), self.d_packed.at(base + q * k * L), 8 * k * L, ctx.ptr)
                        ops.append(td.P2POp(td.isend, send, q))
                        staged.append(send)
                for q, chunks_q in peer_chunks.items():
                    if step >= len(chunks_q):
                        continue
                    d0, d1 = chunks_q[step]
                    qlo, cnt = self.bounds[q][0], (d1 - d0) * L
                    if on_device:  # straight into the leaf block of rank q's columns
                        recv = device_tensor(torch, self.d_leaves.at((qlo + d0) * L), cnt, g.device_index)
                    else:
                        recv = self._host_tensor(("r", q, d0), cnt)
                        staged.append((q, qlo + d0, recv, cnt))
                    ops.append(td.P2POp(td.irecv, recv, q))
                if ops:
                    reqs += td.batch_isend_irecv(ops)
            for req in reqs:
                req.wait()
            if on_device:
                torch.cuda.synchronize()
            else:
                for item in staged:
                    if isinstance(item, tuple):
                        _, col, recv, cnt = item
                        if dry:  # into the tensor the RCCL route receives into: the leaf block of that rank's columns
                            device_tensor(torch, self.d_leaves.at(col * L), cnt, g.device_index).copy_(recv)
                        else:
                            _lib.call("gl_memcpy_h2d", self.d_leaves.at(col * L), recv.data_ptr(), 8 * cnt, ctx.ptr)
                if dry:
                    torch.cuda.synchronize()
            ctx.synchronize()
        _lib.call("gl_merkle_tree_from_columns", self.d_leaves.ptr, self.total_cols, L, L, self.local_cap_height, self.d_digests.ptr,
                  self.d_cap.ptr, ctx.ptr)
        my_cap = self.d_cap.download(0, 4 << self.local_cap_height).reshape(-1, 4)
        cap = np.concatenate(g.gather_caps(my_cap), axis=0)
        return ShardedCommit(d_coeffs=d_values, d_lde=self.d_lde, d_leaves=self.d_leaves, d_digests=self.d_digests, cap=cap, col_lo=self.col_lo,
                             col_hi=self.col_hi, total_cols=self.total_cols, leaf_lo=r * L, leaves_per_rank=L, digest_lo=r * self.n_dig,
                             num_digests=self.n_dig, local_cap_height=self.local_cap_height)

    def _host_tensor(self, key, n_elems):
        t = self._host.get(key)
        if t is None or t.numel() != n_elems:
            t = self.group.torch.empty(n_elems, dtype=self.group.torch.int64)
            self._host[key] = t
        return t


def sharded_commit_from_values(group, ctx, d_values, col_lo, col_hi, total_cols, log_n, rate_bits, cap_height, plan=None):
    """PolynomialBatch::from_values (fri/oracle.rs:709-731) for ONE trace whose columns [col_lo, col_hi) live on this
    rank (`d_values`: [col_hi - col_lo][2^log_n], transformed in place). Per rank: inverse NTT and coset LDE of its
    own columns; the path's ONE exchange — every rank sends, for each of its columns, the leaf range each other
    rank owns (leaf ranges are contiguous because cap subtrees are), point to point, chunk by chunk under the remaining
    LDE; then leaf hashing and the subtrees of its own leaf range; finally an all-gather of the 2^cap_height x 32 B cap.
    With backend "nccl" the exchange runs between device buffers over RCCL; with "gloo" it is staged through host memory
    (how it is tested with ranks sharing one GPU). `plan`: a ShardedCommitPlan of this shape to reuse (its buffers hold
    the result); without one, a plan is made for this call and its buffers belong to the returned ShardedCommit."""
    if plan is None:
        plan = ShardedCommitPlan(group, ctx, total_cols, log_n, rate_bits, cap_height)
    if (plan.col_lo, plan.col_hi) != (col_lo, col_hi):
        raise ValueError("columns must be sharded contiguously by shard_range(total_cols, world, rank)")
    return plan.commit(d_values)


def sharded_open_batch(group, ctx, sc, indices):
    """MerkleTree::get + prove (hash/merkle_tree.rs:392-440) for GLOBAL leaf indices of a ShardedCommit — what
    fri_prover_query_round asks of a tree. Each index is answered by the rank that owns its leaf range: the path
    inside that rank's subtrees is the whole path (log2(N) - cap_height siblings either way). The answers are
    combined with a sum all-reduce (the other ranks contribute zeros), so every rank returns
    (leaves [count, total_cols], siblings [count, layers, 4]) for all indices."""
    from . import _lib

    idx = np.ascontiguousarray(indices, dtype=np.uint64)
    L = sc.leaves_per_rank
    layers = (L.bit_length() - 1) - sc.local_cap_height
    leaves = np.zeros((idx.size, sc.total_cols), dtype=np.uint64)
    sib = np.zeros((idx.size, layers, 4), dtype=np.uint64)
    own = np.flatnonzero((idx >= sc.leaf_lo) & (idx < sc.leaf_lo + L))
    if own.size:
        local = np.ascontiguousarray(idx[own] - np.uint64(sc.leaf_lo))
        lv = np.empty((own.size, sc.total_cols), dtype=np.uint64)
        sb = np.empty((own.size, layers, 4), dtype=np.uint64)
        _lib.call("gl_merkle_open_batch", sc.d_leaves.ptr, 1, L, sc.total_cols, L, sc.local_cap_height,
                  sc.d_digests.ptr if layers else None, local, local.size, lv, sb if layers else None, ctx.ptr)
        leaves[own] = lv
        if layers:
            sib[own] = sb
    if group.td:
        for arr in (leaves, sib):
            if arr.size:
                t = group.torch.from_numpy(arr.view(np.int64).reshape(-1).copy()).to(group._dev())
                group.td.all_reduce(t, op=group.td.ReduceOp.SUM)
                arr[...] = t.cpu().numpy().view(np.uint64).reshape(arr.shape)
    return leaves, sib
