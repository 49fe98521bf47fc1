"""Where the device idles during a proof. Two modes:
  python3 tools/prove_timeline.py run [degree_bits=18] [proofs=6]     (under rocprofv3 --kernel-trace --memory-copy-trace)
      two warm-up proofs, a 0.4 s pause, then `proofs` proofs back to back on one context
  python3 tools/prove_timeline.py analyse <dir with *_kernel_trace.csv [*_memory_copy_trace.csv]> [proofs=6]
      everything after the pause: wall time, union of busy intervals (kernels and copies), launches and copies per proof,
      and the idle gaps grouped by (what ended before the gap -> what started after it)."""
import csv
import glob
import json
import os
import sys
import time


def run(degree_bits, proofs):
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import numpy as np

    import plonky2_gpu_amd as pg
    import synth_circuit
    from plonky2_gpu_amd.challenger import hash_no_pad

    ctx = pg.Context(0)
    circuit, wires, pis = synth_circuit.make(degree_bits, num_wires=234, num_routed=80, num_constants=8, seed=1, gate_table="ed25519")
    synth_circuit.set_public_input_row(wires, hash_no_pad(ctx, pis))
    nc = pg.NativeCircuit(ctx, dict(circuit, circuit_digest=None))
    d_wires = pg.DeviceBuffer.from_host(ctx, np.ascontiguousarray(wires))
    for _ in range(2):
        nc.prove_bytes(d_wires, pis)
    ctx.synchronize()
    time.sleep(0.4)
    t = time.perf_counter()
    for _ in range(proofs):
        nc.prove_bytes(d_wires, pis)
    ctx.synchronize()
    print(json.dumps({"proofs": proofs, "ms_per_proof_wall_clock_under_the_profiler": (time.perf_counter() - t) / proofs * 1e3}))


def short(name):
    name = name.replace("plonky2_hip::", "").replace("(anonymous namespace)::", "").replace("nttk::", "")
    if name.startswith("void "):
        name = name[5:]
    return name.split("(")[0][:48]


def analyse(d, proofs):
    recs = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            recs.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), "k"))
    for f in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            recs.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r.get("Direction", r.get("Name", "?")).replace("MEMORY_COPY_", ""), "c"))
    recs.sort()
    cut, end = 0, recs[0][1]
    for i in range(1, len(recs)):  # the LAST pause of more than 0.3 s: the one run() makes in front of the timed proofs
        if recs[i][0] - end > 300e6:
            cut = i
        end = max(end, recs[i][1])
    recs = recs[cut:]
    t0, t1 = recs[0][0], max(r[1] for r in recs)
    busy, gaps, end, last = 0, {}, recs[0][0], "start"
    for s, e, n, _ in recs:
        if s > end:
            g = gaps.setdefault((last, n), [0, 0])
            g[0] += 1
            g[1] += s - end
            busy += e - s
            end, last = e, n
        elif e > end:
            busy += e - end
            end, last = e, n
    wall = t1 - t0
    out = {"proofs": proofs, "wall_ms_per_proof": wall / proofs / 1e6, "device_busy_ms_per_proof": busy / proofs / 1e6,
           "device_idle_ms_per_proof": (wall - busy) / proofs / 1e6,
           "kernel_launches_per_proof": sum(1 for r in recs if r[3] == "k") / proofs,
           "copies_per_proof": sum(1 for r in recs if r[3] == "c") / proofs}
    print(json.dumps(out))
    print("idle gaps by (before -> after), per proof: count, total us, average us")
    for (a, b), (c, ns) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:40]:
        print("  %-44s -> %-44s %6.1f %9.1f %8.1f" % (a[:44], b[:44], c / proofs, ns / proofs / 1e3, ns / c / 1e3))
    dur = {}
    for s_, e_, n, _ in recs:
        x = dur.setdefault(n, [0, 0])
        x[0] += 1
        x[1] += e_ - s_
    print("durations by kernel / copy, per proof: count, total us, average us (kernels on different streams overlap: sums exceed the wall)")
    for n, (c, ns) in sorted(dur.items(), key=lambda kv: -kv[1][1])[:45]:
        print("  %-48s %6.1f %9.1f %8.1f" % (n[:48], c / proofs, ns / proofs / 1e3, ns / c / 1e3))
    by_after = {}
    for (a, b), (c, ns) in gaps.items():
        x = by_after.setdefault(b, [0, 0])
        x[0] += c
        x[1] += ns
    print("idle gaps by what started after them, per proof: count, total us")
    for b, (c, ns) in sorted(by_after.items(), key=lambda kv: -kv[1][1])[:25]:
        print("  %-48s %6.1f %9.1f" % (b[:48], c / proofs, ns / proofs / 1e3))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(int(sys.argv[2]) if len(sys.argv) > 2 else 18, int(sys.argv[3]) if len(sys.argv) > 3 else 6)
    else:
        analyse(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 6)
