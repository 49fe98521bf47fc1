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


def run_in_flight(degree_bits, proofs, k):
    """`k` host threads x own context x own circuit handle, `proofs` proofs each, after a warm-up and a 0.4 s pause (for `overlap`)"""
    import threading

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import numpy as np

    import plonky2_gpu_amd as pg
    import synth_circuit
    from plonky2_gpu_amd.challenger import hash_no_pad

    ctxs = [pg.Context(0) for _ in range(k)]
    circuit, wires, pis = synth_circuit.make(degree_bits, num_wires=234, num_routed=80, num_constants=8, seed=1, gate_table="ed25519")
    synth_circuit.set_public_input_row(wires, hash_no_pad(ctxs[0], pis))
    ncs = [pg.NativeCircuit(c, dict(circuit, circuit_digest=None)) for c in ctxs]
    bufs = [pg.DeviceBuffer.from_host(c, np.ascontiguousarray(wires)) for c in ctxs]
    for i in range(k):
        for _ in range(2):
            ncs[i].prove_bytes(bufs[i], pis)
        ctxs[i].synchronize()
    time.sleep(0.4)

    def work(i):
        for _ in range(proofs):
            ncs[i].prove_bytes(bufs[i], pis)
        ctxs[i].synchronize()

    t = time.perf_counter()
    threads = [threading.Thread(target=work, args=(i,)) for i in range(k)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    print(json.dumps({"in_flight": k, "proofs": proofs * k, "ms_per_proof_wall_clock_under_the_profiler": (time.perf_counter() - t) / (proofs * k) * 1e3}))


def overlap(d, proofs):
    """Everything after the pause of a run_in_flight trace: per stream, the time its kernels run; the time kernels of at least two
    different streams run at once; which kernels of one proof run beside which of the other."""
    recs = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            recs.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), int(r["Stream_Id"]), int(r["Thread_Id"])))
    recs.sort()
    cut, end = 0, recs[0][1]
    for i in range(1, len(recs)):
        if recs[i][0] - end > 300e6:
            cut = i
        end = max(end, recs[i][1])
    recs = recs[cut:]
    t0, t1 = recs[0][0], max(r[1] for r in recs)
    threads = sorted({r[4] for r in recs})
    ev = []
    for s_, e_, n, st, th in recs:
        ev.append((s_, 1, th, n))
        ev.append((e_, -1, th, n))
    ev.sort()
    active = {th: {} for th in threads}   # thread -> kernel name -> count running
    last, any_busy, both_busy = t0, 0, 0
    pair_ns = {}
    for t, dlt, th, n in ev:
        running = [x for x in threads if active[x]]
        if running:
            any_busy += t - last
        if len(running) >= 2:
            both_busy += t - last
            a, b = (max(active[x], key=lambda k_: active[x][k_]) for x in running[:2])
            key = tuple(sorted((a, b)))
            pair_ns[key] = pair_ns.get(key, 0) + (t - last)
        last = t
        cnt = active[th].get(n, 0) + dlt
        if cnt:
            active[th][n] = cnt
        else:
            active[th].pop(n, None)
    per_thread = {}
    for th in threads:
        iv = sorted((r[0], r[1]) for r in recs if r[4] == th)
        busy, e = 0, iv[0][0]
        for s_, e_ in iv:
            if s_ > e:
                busy += e_ - s_
                e = e_
            elif e_ > e:
                busy += e_ - e
                e = e_
        per_thread[str(th)] = {"kernels": len(iv), "streams": len({r[3] for r in recs if r[4] == th}), "busy_ms": busy / 1e6}
    print(json.dumps({"host_threads": len(threads), "proofs": proofs, "wall_ms": (t1 - t0) / 1e6, "ms_per_proof": (t1 - t0) / 1e6 / proofs,
                      "device_busy_ms": any_busy / 1e6, "kernels_of_two_host_threads_running_at_once_ms": both_busy / 1e6,
                      "frac_of_wall_with_both": both_busy / (t1 - t0), "per_host_thread": per_thread}))
    print("what runs beside what (kernel of one proof | kernel of the other), ms")
    for (a, b), ns in sorted(pair_ns.items(), key=lambda kv: -kv[1])[:25]:
        print("  %-44s | %-44s %9.2f" % (a[:44], b[:44], ns / 1e6))


if __name__ == "__main__":
    if sys.argv[1] == "run_in_flight":
        run_in_flight(int(sys.argv[2]) if len(sys.argv) > 2 else 18, int(sys.argv[3]) if len(sys.argv) > 3 else 6, int(sys.argv[4]) if len(sys.argv) > 4 else 2)
    elif sys.argv[1] == "overlap":
        overlap(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 12)
    elif sys.argv[1] == "run":
        run(int(sys.argv[2]) if len(sys.argv) > 2 else 18, int(sys.argv[3]) if len(sys.argv) > 3 else 6)
    else:
        analyse(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 6)
