"""Soak run for the concurrent contexts of round 6 (not part of the suite): K host threads, each with its own context, hammer
different workloads AT THE SAME TIME for some minutes — whole proofs (one thread with its own circuit handle, two threads sharing
one), pipelined commits with the leaf-major copy, natural-order NTTs (which stage through the context's workspace), the
device-resident challenger — and every result is compared with what the same call gives alone. A missing dependency between
streams, a buffer handed to the wrong context or shared state that should have been per context shows up as a rare mismatch.
    python tests/soak_contexts.py [minutes=3] [threads=4]
Prints a JSON summary; exits non-zero on the first mismatch."""
import hashlib
import json
import os
import sys
import threading
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import plonky2_gpu_amd as pg  # noqa: E402
from oracle import accel, oracle as o, prove_ref, serialize_ref  # noqa: E402
from plonk_instance import make_full_circuit  # noqa: E402
from plonky2_gpu_amd import _lib  # noqa: E402


def sha(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def main():
    minutes = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
    n_threads = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    ctxs = [pg.Context(0) for _ in range(n_threads)]
    with accel.c_backend():
        circuit, wires, pis = make_full_circuit(10, seed=5, arity_bits=(4, 4), cap_height=4, num_queries=28)
        want = serialize_ref.proof_bytes(prove_ref.prove(circuit, wires, pis))
    shared = pg.NativeCircuit(ctxs[0], dict(circuit, circuit_digest=None), compile_gates=True)
    flat = np.ascontiguousarray(np.array(wires, dtype=np.uint64).reshape(-1))
    vals = o.random_field((135, 1 << 14), seed=77)
    exp = o.commit_from_values(vals, 3, 4, threads=8)
    x20 = o.random_field((8, 1 << 20), seed=78)
    stop_at = time.time() + 60 * minutes
    counts, errors = [dict() for _ in range(n_threads)], []

    def prove_loop(i, nc, own):
        d_w = pg.DeviceBuffer.from_host(ctxs[i], flat)
        while time.time() < stop_at and not errors:
            if nc.prove_bytes(d_w, pis, ctx=ctxs[i]) != want:
                errors.append("thread %d: proof differs (%s circuit handle)" % (i, "own" if own else "shared"))
            counts[i]["proofs"] = counts[i].get("proofs", 0) + 1
        d_w.free()

    def commit_loop(i):
        ctx = ctxs[i]
        n, n_ext, cols = 1 << 14, 1 << 17, 135
        d_vals = pg.DeviceBuffer.from_host(ctx, vals)
        d_work, d_lde, d_leaves = pg.DeviceBuffer(ctx, cols * n), pg.DeviceBuffer(ctx, cols * n_ext), pg.DeviceBuffer(ctx, cols * n_ext)
        d_dig, d_cap = pg.DeviceBuffer(ctx, 8 * (n_ext - 16)), pg.DeviceBuffer(ctx, 64)
        first = None
        while time.time() < stop_at and not errors:
            _lib.call("gl_memcpy_d2d", d_work.ptr, d_vals.ptr, cols * n * 8, ctx.ptr)
            _lib.call("gl_commit_from_values", d_work.ptr, cols, 14, 3, 4, 0, 7, d_lde.ptr, d_leaves.ptr, d_dig.ptr, d_cap.ptr, ctx.ptr)
            got = sha(d_cap.download(), d_dig.download(), d_leaves.download(0, 1 << 20))
            if first is None:
                first = got
                if not (d_cap.download().reshape(16, 4) == o.canon(exp["cap"])).all():
                    errors.append("thread %d: the commit's cap differs from the oracle's" % i)
            elif got != first:
                errors.append("thread %d: commit differs from its first result" % i)
            counts[i]["commits"] = counts[i].get("commits", 0) + 1

    def ntt_loop(i):
        ctx = ctxs[i]
        buf = pg.DeviceBuffer.from_host(ctx, x20)
        first = None
        d_ch, d_out = pg.DeviceBuffer(ctx, 32), pg.DeviceBuffer(ctx, 8)
        import ctypes
        src = (_lib.GlObserveSrc * 1)(_lib.GlObserveSrc(buf.ptr, 200, 0))
        while time.time() < stop_at and not errors:
            _lib.call("gl_ntt_batch", buf.ptr, 8, 20, 1 << 20, 0, 0, ctx.ptr)   # natural order: through the context's workspace
            f = sha(buf.download(0, 1 << 18))
            _lib.call("gl_challenger_step", d_ch.ptr, ctypes.addressof(src), 1, 8, d_out.ptr, 1, ctx.ptr)
            c = sha(d_out.download())
            _lib.call("gl_ntt_batch", buf.ptr, 8, 20, 1 << 20, 1, 0, ctx.ptr)
            back = buf.download(0, 1 << 18)
            if not (back == o.canon(x20).reshape(-1)[: 1 << 18]).all():
                errors.append("thread %d: ifft(fft(x)) != x" % i)
            if first is None:
                first = (f, c)
            elif (f, c) != first:
                errors.append("thread %d: transform or challenges differ from their first results" % i)
            counts[i]["ntt_round_trips"] = counts[i].get("ntt_round_trips", 0) + 1

    own = pg.NativeCircuit(ctxs[-1], dict(circuit, circuit_digest=None), compile_gates=True) if n_threads >= 4 else None
    jobs = []
    for i in range(n_threads):
        kind = i % 4
        if kind == 0:
            jobs.append(threading.Thread(target=prove_loop, args=(i, shared, False)))
        elif kind == 1:
            jobs.append(threading.Thread(target=commit_loop, args=(i,)))
        elif kind == 2:
            jobs.append(threading.Thread(target=ntt_loop, args=(i,)))
        else:
            jobs.append(threading.Thread(target=prove_loop, args=(i, own if i == n_threads - 1 and own else shared, i == n_threads - 1 and own is not None)))
    t0 = time.time()
    for t in jobs:
        t.start()
    for t in jobs:
        t.join()
    print(json.dumps({"minutes": round((time.time() - t0) / 60, 2), "threads": n_threads, "per_thread": counts, "errors": errors}))
    sys.exit(1 if errors else 0)


if __name__ == "__main__":
    main()
