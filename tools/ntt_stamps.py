#!/usr/bin/env python3
"""Where a wavefront of the NTT pass kernels spends its cycles: runs the 2^20 batch transform on a DIAGNOSTIC build of the
library (csrc/ntt.hip compiled with -DPLONKY2_NTT_STAMPS: s_memtime stamps around each phase of the tile loop, totals added
per pass kind) and prints, per pass, the share of each phase. Build the diagnostic library with

    hipcc -O3 -std=c++17 -fPIC -fvisibility=hidden --offload-arch=gfx950 -DPLONKY2_NTT_STAMPS -c plonky2_gpu_amd/csrc/ntt.hip -o /tmp/ntt_stamps.o
    hipcc -shared -fPIC --offload-arch=gfx950 /tmp/ntt_stamps.o plonky2_gpu_amd/csrc/build/{merkle,plonk,fri,gate_jit,prove,capi}.o -lhiprtc -o <dir>/libplonky2_hip.so

and run   python tools/ntt_stamps.py <dir>/libplonky2_hip.so out.jsonl   (never the product library: stamping serialises phases)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PHASES = ["prologue", "radix rounds", "barrier + results LDS->regs", "barrier (results read)", "next tile: wait loads, regs->LDS, barrier",
          "twiddle-chain look-ups", "stores issued", "loads issued"]

if len(sys.argv) >= 4 and sys.argv[3] == "--child":
    import numpy as np

    from plonky2_gpu_amd import _lib

    _lib.LIB_PATH = sys.argv[1]
    import plonky2_gpu_amd as pg

    ctx = pg.Context(0)
    n, batch = 1 << 20, 64
    buf = pg.DeviceBuffer.from_host(ctx, np.random.default_rng(1).integers(0, 0xFFFFFFFF00000001, size=(batch, n), dtype=np.uint64))
    for order in (0, 1):
        for _ in range(4):
            _lib.call("gl_ntt_batch", buf.ptr, batch, 20, n, 0, order, ctx.ptr)
    ctx.synchronize()
    sys.exit(0)

lib, out = sys.argv[1], sys.argv[2]
env = dict(os.environ, PLONKY2_NTT_STAMPS_OUT=out)
subprocess.check_call([sys.executable, os.path.abspath(__file__), lib, out, "--child"], env=env)
for line in open(out):
    d = json.loads(line)
    tot = sum(d["c%d" % k] for k in range(8))
    if not d["c8"]:
        continue
    print(f"{d['pass']}: {d['c8']} wave-tiles, {tot / d['c8']:.0f} cycles per wave and tile")
    for k, name in enumerate(PHASES):
        print(f"    {name:48s} {d['c%d' % k] / d['c8']:9.0f} cycles  {100.0 * d['c%d' % k] / tot:5.1f} %")
