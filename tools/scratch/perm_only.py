import sys
sys.path.insert(0,'/root/repo')
import numpy as np, plonky2_gpu_amd as pg, time
from plonky2_gpu_amd import _lib
ctx=pg.Context(0)
n=1<<22
rng=np.random.default_rng(1)
buf=pg.DeviceBuffer.from_host(ctx,rng.integers(0,pg.P,size=12*n,dtype=np.uint64))
for _ in range(3):
    ctx.synchronize(); t=time.perf_counter()
    _lib.call("gl_poseidon_permute_batch",buf.ptr,n,ctx.ptr); ctx.synchronize()
    dt=time.perf_counter()-t
print("perm/s %.3e"%(n/dt))
