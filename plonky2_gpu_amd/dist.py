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
    def __init__(self, backend=None):
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.td = None
        self.backend = None
        if self.world > 1:
            import torch
            import torch.distributed as td

            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
            self.backend = backend or "gloo"
            if self.backend == "nccl":
                torch.cuda.set_device(self.local_rank)
            td.init_process_group(backend=self.backend, rank=self.rank, world_size=self.world)
            self.td, self.torch = td, torch

    def _dev(self):
        return self.torch.device("cuda", self.local_rank) if self.backend == "nccl" else self.torch.device("cpu")

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
