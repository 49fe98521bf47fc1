"""Proofs per second of ONE GPU with K proofs in flight (configs[4]'s per-GPU factor): K host threads, each with its own context
(its own streams, workspace, hashing stream: csrc/capi.hip CtxState) and — by default — its own circuit handle, prove the
ed25519-shaped synthetic circuit of tools/bench_prove.py back to back. A proof has latency-bound phases (the transcript's serial
sponge, tree layers below 2^16 nodes, openings, host round trips) during which one proof alone leaves the chip idle; a second proof
in flight fills them. Every proof's bytes are compared with the bytes the same witness gives alone on one context.
usage: python tools/bench_inflight.py [degree_bits=18] [reps=8] [ks=1,2,3] [share_circuit=0] [num_wires=234]
share_circuit=1: the threads prove with ONE circuit handle (one compiled gate kernel: its launches take turns, gate_jit.hip)."""
import json
import os
import sys
import threading
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np  # noqa: E402

import plonky2_gpu_amd as pg  # noqa: E402
import synth_circuit  # noqa: E402
from plonky2_gpu_amd.challenger import hash_no_pad  # noqa: E402


def run(degree_bits=18, reps=8, ks=(1, 2, 3), share=False, num_wires=234, device=0):
    table = "ed25519" if num_wires == 234 else "mini"
    kmax = max(ks)
    ctxs = [pg.Context(device) for _ in range(kmax)]
    circuit, wires, pis = synth_circuit.make(degree_bits, num_wires=num_wires, num_routed=80, num_constants=8, seed=1, gate_table=table)
    synth_circuit.set_public_input_row(wires, hash_no_pad(ctxs[0], pis))
    wires = np.ascontiguousarray(wires)
    ncs, d_wires = [], []
    for i, c in enumerate(ctxs):
        ncs.append(ncs[0] if (share and i) else pg.NativeCircuit(c, dict(circuit, circuit_digest=None)))
        d_wires.append(pg.DeviceBuffer.from_host(c, wires))
    expect = ncs[0].prove_bytes(d_wires[0], pis)
    for i, c in enumerate(ctxs):  # warm-up of every context: buffer pool, tables, hashing stream
        assert ncs[i].prove_bytes(d_wires[i], pis, ctx=c) == expect
        c.synchronize()
    out = {"workload": f"prove() of the ed25519-shaped synthetic circuit (n=2^{degree_bits}, {num_wires} wires, gate_table={table}), "
                       f"{reps} proofs per thread, witness resident", "share_circuit": bool(share), "k": {}}
    for k in ks:
        bad, done_at = [], [0.0] * k

        def work(i):
            for _ in range(reps):
                if ncs[i].prove_bytes(d_wires[i], pis, ctx=ctxs[i]) != expect:
                    bad.append(i)
            ctxs[i].synchronize()
            done_at[i] = time.perf_counter()

        best = None
        for attempt in range(2):
            threads = [threading.Thread(target=work, args=(i,)) for i in range(k)]
            for c in ctxs:
                c.synchronize()
            t0 = time.perf_counter()
            for t in threads:
                t.start()
            for t in threads:
                t.join()
            dt = max(done_at) - t0
            best = dt if best is None else min(best, dt)
        if bad:
            raise SystemExit(f"bench_inflight: proofs of thread(s) {sorted(set(bad))} differ from the single-context proof at k={k}")
        out["k"][str(k)] = {"proofs_per_s": round(k * reps / best, 3), "ms_per_proof": round(best / (k * reps) * 1e3, 3),
                            "wall_s": round(best, 4)}
    base = out["k"].get("1")
    if base:
        for k, v in out["k"].items():
            v["vs_one_in_flight"] = round(v["proofs_per_s"] / base["proofs_per_s"], 4)
    for b in d_wires:
        b.free()
    for nc in set(ncs):
        nc.close()
    for c in ctxs:
        c.close()
    return out


if __name__ == "__main__":
    a = sys.argv[1:]
    print(json.dumps(run(int(a[0]) if len(a) > 0 else 18, int(a[1]) if len(a) > 1 else 8,
                         tuple(int(x) for x in (a[2] if len(a) > 2 else "1,2,3").split(",")),
                         bool(int(a[3])) if len(a) > 3 else False, int(a[4]) if len(a) > 4 else 234)))
