"""Soak run (not part of the suite): the same commit and the same proof over and over, every result compared with the first.
The pipelined commit and gl_prove use several streams and events; a missing dependency would show up as a rare mismatch.
    python tests/soak.py [minutes=5]
Shapes: the pipelined commit at 135 x 2^14 (rate 8, cap 4) from_values and from_coeffs with the leaf-major copy; the 13-gate circuit
at 2^10 rows through gl_prove (compiled gates) — whose first proof is also checked against the oracle prover; a natural-order and a
bit-reversed batch NTT at 2^20 and 2^21 against their first results. Prints a JSON summary; exits non-zero on the first mismatch."""
import hashlib
import json
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import plonky2_gpu_amd as pg  # noqa: E402
from oracle import accel, oracle as o, prove_ref, serialize_ref  # noqa: E402
from plonk_instance import make_full_circuit  # noqa: E402
from plonky2_gpu_amd import _lib  # noqa: E402


def digest(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def main():
    minutes = float(sys.argv[1]) if len(sys.argv) > 1 else 5.0
    ctx = pg.Context(0)
    vals = o.random_field((135, 1 << 14), seed=77)
    exp = o.commit_from_values(vals, 3, 4, threads=8)
    first = {}
    with accel.c_backend():
        circuit, wires, pis = make_full_circuit(10, seed=5, arity_bits=(4, 4), cap_height=4, num_queries=28)
        want = serialize_ref.proof_bytes(prove_ref.prove(circuit, wires, pis))
    nc = pg.NativeCircuit(ctx, dict(circuit, circuit_digest=None), compile_gates=True)
    ntt_in = {lg: o.random_field(((1 << 24) >> lg, 1 << lg), seed=lg) for lg in (20, 21)}
    bufs = {lg: pg.DeviceBuffer.from_host(ctx, a) for lg, a in ntt_in.items()}
    counts = dict(commit=0, proof=0, ntt=0)
    t_end = time.time() + 60 * minutes
    while time.time() < t_end:
        for fv in (True, False):
            src = vals if fv else o.canon(exp["coeffs"])
            b = (pg.PolynomialBatch.from_values if fv else pg.PolynomialBatch.from_coeffs)(ctx, src, 3, False, 4, leaf_major=True)
            d = digest(b.merkle_tree.cap, b.merkle_tree.digests, b.merkle_tree.d_leaves.download())
            if counts["commit"] < 2:
                assert (b.merkle_tree.cap == o.canon(exp["cap"])).all() and (b.merkle_tree.digests == o.canon(exp["digests"]).reshape(-1, 4)).all()
            if first.setdefault(("commit", fv), d) != d:
                print(json.dumps(dict(mismatch="commit", from_values=fv, after=counts)))
                sys.exit(1)
            counts["commit"] += 1
        for _ in range(3):
            data = nc.prove_bytes(wires, pis)
            if data != want:
                print(json.dumps(dict(mismatch="proof", after=counts)))
                sys.exit(1)
            counts["proof"] += 1
        for lg, buf in bufs.items():
            for order in (0, 1):
                buf.upload(ntt_in[lg])
                _lib.call("gl_ntt_batch", buf.ptr, ntt_in[lg].shape[0], lg, 1 << lg, 0, order, ctx.ptr)
                d = digest(buf.download())
                if first.setdefault(("ntt", lg, order), d) != d:
                    print(json.dumps(dict(mismatch="ntt", log_n=lg, order=order, after=counts)))
                    sys.exit(1)
                counts["ntt"] += 1
    nc.close()
    print(json.dumps(dict(minutes=minutes, all_identical=True, **counts)))


if __name__ == "__main__":
    main()
