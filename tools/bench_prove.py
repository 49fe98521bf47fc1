"""Time prove() on the device for a synthetic circuit of the ed25519 proof's SHAPE (BASELINE.json
config 4: n = 2^18, 234 wires / 80 routed, 88 preprocessed polynomials, 2 challenges, rate 8,
cap_height 4, FRI arities [4,4,4,4], 28 queries, 16 PoW bits). A proof of exactly this shape is checked
by the oracle's verifier in tests/test_gpu_prove.py (tools/ do not use oracle/); verify=1 here only checks
the wire-format round trip. The ed25519 circuit itself needs the Rust toolchain: the ROWS here use
Noop/Constant/PublicInput/Arithmetic{20} only, but the circuit's gate LIST is the real one by default, and the list is
what the quotient stage's cost depends on — every stage runs at the real shape and cost.
usage: python tools/bench_prove.py [degree_bits=18] [num_wires=234] [reps=3] [verify=1] [native=1] [gate_table=ed25519|mini]
gate_table=ed25519 (default at 234 wires): the circuit declares the ed25519 circuit's whole 25-gate table, so the quotient
stage costs what the real circuit's does (tools/synth_circuit.py); mini: only the four instantiated kinds.
native=1 times gl_prove (the library's own C++ prover, csrc/prove.hip); native=0 the Python host mirror."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np  # noqa: E402

import plonky2_gpu_amd as pg  # noqa: E402
import synth_circuit  # noqa: E402
from plonky2_gpu_amd.challenger import hash_no_pad  # noqa: E402


def main():
    degree_bits = int(sys.argv[1]) if len(sys.argv) > 1 else 18
    num_wires = int(sys.argv[2]) if len(sys.argv) > 2 else 234
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    verify = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    native = int(sys.argv[5]) if len(sys.argv) > 5 else 1
    table = sys.argv[6] if len(sys.argv) > 6 else ("ed25519" if num_wires == 234 else "mini")
    ctx = pg.Context(0)
    t = time.perf_counter()
    circuit, wires, pis = synth_circuit.make(degree_bits, num_wires=num_wires, num_routed=80, num_constants=8, seed=1, gate_table=table)
    synth_circuit.set_public_input_row(wires, hash_no_pad(ctx, pis))
    gen_s = time.perf_counter() - t
    t = time.perf_counter()
    cd = pg.CircuitData(ctx, dict(circuit, circuit_digest=[0, 0, 0, 0]))
    cap = cd.constants_sigmas_commitment.merkle_tree.cap.tolist()
    flat = [x for h in cap for x in h]
    pad = [1] + [0] * 10 + [1]  # hash_pad of the empty domain separator (plonk/config.rs:44-52)
    cd.circuit_digest = hash_no_pad(ctx, flat + hash_no_pad(ctx, pad) + [degree_bits])  # circuit_builder.rs:915-927
    ctx.synchronize()
    build_s = time.perf_counter() - t
    wires = np.ascontiguousarray(wires)
    d_wires = pg.DeviceBuffer.from_host(ctx, wires)
    # what a host-resident witness adds to every proof: one H2D of the [num_wires][n] matrix (pinned staging)
    staging = pg.PinnedArray(wires.size)
    staging.array[:] = wires.reshape(-1)
    h2d = []
    for _ in range(3):
        ctx.synchronize()
        t = time.perf_counter()
        d_wires.upload(staging.array)
        ctx.synchronize()
        h2d.append((time.perf_counter() - t) * 1e3)
    nc = pg.NativeCircuit(ctx, dict(circuit, circuit_digest=cd.circuit_digest)) if native else None
    runs, untimed = [], []
    for r in range(reps + 1):
        timing = {}
        ctx.synchronize()
        t = time.perf_counter()
        if native:
            data = nc.prove_bytes(d_wires, pis, timing)
        else:
            proof = pg.prove(ctx, cd, d_wires, pis, timing)
        ctx.synchronize()
        timing["total"] = (time.perf_counter() - t) * 1e3
        if r:  # first run warms up (table builds, allocator)
            runs.append(timing)
    for r in range(reps):  # without the per-stage synchronisations
        ctx.synchronize()
        t = time.perf_counter()
        if native:
            data = nc.prove_bytes(d_wires, pis)
        else:
            proof = pg.prove(ctx, cd, d_wires, pis)
        ctx.synchronize()
        untimed.append((time.perf_counter() - t) * 1e3)
    pipelined = None
    if native:
        # A stream of proofs whose witnesses arrive in (pinned) host memory: the upload of witness k+1 is queued on
        # the second stream while gl_prove works on witness k (two device buffers, swapped each proof)
        d_next = pg.DeviceBuffer(ctx, wires.size)
        bufs, period = [d_wires, d_next], []
        bufs[0].upload(staging.array)
        ctx.synchronize()
        for r in range(reps + 2):
            t = time.perf_counter()
            bufs[1].upload_async(staging)
            piped = nc.prove_bytes(bufs[0], pis)
            ctx.synchronize()
            period.append((time.perf_counter() - t) * 1e3)
            bufs.reverse()
            if piped != data:
                raise SystemExit("bench_prove: the proof from the double-buffered witness differs")
        pipelined = round(min(period[1:]), 3)
        d_next.free()
    staging.free()
    if native:
        proof = pg.serialization.proof_from_bytes(data, circuit)
    best = min(runs, key=lambda d: d["total"])
    out = dict(workload=f"prove() synthetic circuit n=2^{degree_bits} wires={num_wires} routed=80 preprocessed=88 gate_table={table} "
                        f"({len(circuit['gates'])} gates, {circuit['num_gate_constraints']} constraints; rows use noop/const/pi/arith20)",
               reps=reps, witness_gen_s=round(gen_s, 2), circuit_build_s=round(build_s, 2),
               witness_h2d_ms=round(min(h2d), 3), witness_MiB=round(wires.size * 8 / 2**20, 1),
               host_witness_ms_upload_then_prove=round(min(h2d) + min(untimed), 3), host_witness_ms_pipelined=pipelined,
               best_ms={k: round(v, 3) for k, v in best.items()},
               mean_total_ms=round(sum(d["total"] for d in runs) / len(runs), 3), best_ms_without_stage_syncs=round(min(untimed), 3),
               prover="gl_prove (native)" if native else "python host mirror", proof_bytes=len(pg.serialization.proof_to_bytes(proof)),
               pow_witness=proof["opening_proof"]["pow_witness"])
    if verify:
        # oracle-free self-check (tools/ may not use oracle/; the verifier runs in
        # tests/test_gpu_prove.py::test_full_size_proof_is_accepted_by_the_oracle_verifier on this very shape)
        out["wire_format_round_trip"] = bool(native and pg.serialization.proof_to_bytes(proof) == data) if native else None
    print(json.dumps(out))


if __name__ == "__main__":
    main()
