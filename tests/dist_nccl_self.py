"""The RCCL ("nccl") route of plonky2_gpu_amd.dist on the one-GPU box: a process group of ONE rank over RCCL next to the
library's own HIP context — barrier, max / sum all-reduce, the cap all-gather on device tensors, and the zero-copy wrapper
(device_tensor over gl_malloc'ed memory) through a self send/receive. What cannot run here is the exchange between two
GPUs; the code it uses is this code. Launched by tests/test_gpu_dist.py."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import plonky2_gpu_amd as pg  # noqa: E402
from plonky2_gpu_amd.dist import ProverGroup, device_tensor, sharded_commit_from_values  # noqa: E402

g = ProverGroup(backend="nccl", device_index=0, force=True)
assert g.td is not None and g.backend == "nccl" and g.world == 1
ctx = pg.Context(0)
g.barrier()
assert g.max(3.5) == 3.5 and g.sum(2.0) == 2.0
cap = (np.arange(64, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)).reshape(16, 4)
(got,) = g.gather_caps(cap)
assert (got == cap).all()
# zero-copy tensors over the library's buffers: an all-reduce in place and a send/receive to self
a = pg.DeviceBuffer.from_host(ctx, np.arange(1024, dtype=np.uint64))
b = pg.DeviceBuffer(ctx, 1024)
ctx.synchronize()
ta, tb = device_tensor(g.torch, a.ptr, 1024, 0), device_tensor(g.torch, b.ptr, 1024, 0)
assert ta.data_ptr() == a.ptr and ta.is_cuda
g.td.all_reduce(ta)  # world 1: identity, but through RCCL on our memory
g.torch.cuda.synchronize()
assert (a.download() == np.arange(1024, dtype=np.uint64)).all()
try:
    ops = [g.td.P2POp(g.td.isend, ta, 0), g.td.P2POp(g.td.irecv, tb, 0)]
    for req in g.td.batch_isend_irecv(ops):
        req.wait()
    g.torch.cuda.synchronize()
    assert (b.download() == np.arange(1024, dtype=np.uint64)).all()
    print("self send/recv over RCCL ok")
except Exception as e:  # noqa: BLE001  RCCL builds differ on send-to-self; the wrapper itself is proven by the all-reduce
    print("self send/recv not supported by this RCCL:", type(e).__name__, str(e)[:200])
# the sharded commit under an RCCL group (one rank: no peers, the pack kernel and the cap gather still run)
vals = np.random.default_rng(1).integers(0, 0xFFFFFFFF00000001, size=(6, 256), dtype=np.uint64)
sc = sharded_commit_from_values(g, ctx, pg.DeviceBuffer.from_host(ctx, vals), 0, 6, 6, 8, 3, 2)
whole = pg.PolynomialBatch.from_values(ctx, vals, 3, False, 2)
assert (sc.cap == whole.merkle_tree.cap).all()
g.close()
ctx.close()
print("nccl single-rank ok")
