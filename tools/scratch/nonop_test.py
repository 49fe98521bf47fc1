import sys
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, plonky2_gpu_amd as pg
from plonky2_gpu_amd import _lib
ctx=pg.Context(0)
P=pg.P
rng=np.random.default_rng(1)
edge=[0,1,2,P-1,P-2,2**32-1,2**32,2**32+1,2**63,2**64-1,2**64-2**32,2**64-2**32+1, 0xFFFFFFFF00000000, 0xFFFFFFFFFFFFFFFF]
a=np.concatenate([rng.integers(0,2**64,size=1<<22,dtype=np.uint64), np.repeat(np.array(edge,dtype=np.uint64),len(edge))])
b=np.concatenate([rng.integers(0,2**64,size=1<<22,dtype=np.uint64), np.tile(np.array(edge,dtype=np.uint64),len(edge))])
da,db=pg.DeviceBuffer.from_host(ctx,a),pg.DeviceBuffer.from_host(ctx,b)
o1,o2=pg.DeviceBuffer(ctx,a.size),pg.DeviceBuffer(ctx,a.size)
_lib.call("gl_debug_field_op",2,da.ptr,db.ptr,o1.ptr,a.size,ctx.ptr)
_lib.call("gl_debug_field_op",13,da.ptr,db.ptr,o2.ptr,a.size,ctx.ptr)
x,y=o1.download(),o2.download()
c=lambda v: np.where(v>=np.uint64(P), v-np.uint64(P), v)
print("mismatches (canonical):", int((c(x)!=c(y)).sum()), "of", a.size, "raw mismatches", int((x!=y).sum()))
idx=np.flatnonzero(x!=y)[:10]
for i in idx:
    print(hex(int(a[i])),hex(int(b[i])),hex(int(x[i])),hex(int(y[i])), hex((int(a[i])*int(b[i]))%P))
# determinism of each
o3=pg.DeviceBuffer(ctx,a.size)
_lib.call("gl_debug_field_op",13,da.ptr,db.ptr,o3.ptr,a.size,ctx.ptr)
print("op13 run-to-run raw mismatches", int((o3.download()!=y).sum()))
_lib.call("gl_debug_field_op",2,da.ptr,db.ptr,o3.ptr,a.size,ctx.ptr)
print("op2 run-to-run raw mismatches", int((o3.download()!=x).sum()))
# one wave alone on the chip: back-to-back issue from the same wave
bad=0
for it in range(1500):
    off=(it*64)%(a.size-64)
    if it%3==0: off=(1<<22)+ (it*7)%(len(edge)*len(edge)-64)
    _lib.call("gl_debug_field_op",13,da.ptr+8*off,db.ptr+8*off,o3.ptr,64,ctx.ptr)
    got=o3.download(0,64)
    bad+=int((c(got)!=c(x[off:off+64])).sum())
print("single-wave launches: canonical mismatches", bad)
