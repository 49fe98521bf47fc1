"""Differential campaign against the C prover (oracle/prove_oracle.c), which is fast enough for larger and wider cases than
fuzz_prove.py's Python oracle: randomly shaped circuits of three families — the four basic gates (12 wires), every gate kind of the
ed25519 list (135 wires), the recursion-shaped circuit with the eight upstream gate kinds (135 wires) — at 2^3 .. 2^12 rows, random
FRI shapes, rate, cap height, proof-of-work bits, query counts, compiled or interpreted gates, and blinded (gl_prove_zk with random
salts, a tenth of them raw 64-bit words) in a third of the cases. Every proof byte for byte. Not part of the test suite:
    python tests/fuzz_prove_c.py [cases=40] [seed=1]"""
import json
import os
import random
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

import plonky2_gpu_amd as pg  # noqa: E402
from oracle import accel, prove_c  # noqa: E402
from plonk_instance import make_circuit, make_full_circuit, make_recursion_circuit  # noqa: E402

P = 0xFFFFFFFF00000001


def fri_shape(rng, degree_bits, rate_bits):
    cap_height = rng.choice([0, 1, 2, 3, 4])
    cap_height = min(cap_height, degree_bits + rate_bits)
    arity, bits = [], degree_bits
    while bits > 0 and rng.random() < 0.8:
        ab = rng.choice([1, 2, 3, 4, 4])
        if ab > bits or bits + rate_bits - ab < cap_height:
            break
        arity.append(ab)
        bits -= ab
    return cap_height, tuple(arity)


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = random.Random(seed)
    ctx = pg.Context(0)
    t0 = time.time()
    for k in range(cases):
        family = rng.choice(["mini", "mini", "full", "recursion"])
        degree_bits = rng.choice([3, 4, 5, 6, 7, 8, 9, 10, 11, 12] if family == "mini" else [4, 5, 6, 7, 8, 9, 10])
        cap_height, arity = fri_shape(rng, degree_bits, 3)
        kw = dict(arity_bits=arity, cap_height=cap_height, pow_bits=rng.choice([0, 2, 5, 9]), num_queries=rng.choice([1, 3, 7, 28]))
        with accel.c_backend():
            if family == "mini":
                circuit, wires, pis = make_circuit(degree_bits, seed=7000 * seed + k, two_groups=rng.random() < 0.5, num_challenges=rng.choice([1, 2, 2, 3]), **kw)
            elif family == "full":
                circuit, wires, pis = make_full_circuit(degree_bits, seed=7000 * seed + k, **kw)
            else:
                circuit, wires, pis = make_recursion_circuit(degree_bits, seed=7000 * seed + k, **kw)
        salts = None
        if rng.random() < 0.33:
            circuit = dict(circuit, fri_params=dict(circuit["fri_params"], hiding=True))
            nprng = np.random.default_rng(seed * 100003 + k)
            n_ext = 1 << (degree_bits + 3)
            salts = nprng.integers(0, 2**64 if rng.random() < 0.3 else P, size=(3, 4, n_ext), dtype=np.uint64)
        compile_gates = rng.random() < 0.5
        exp = prove_c.prove(circuit, wires, pis, salts=salts)
        nc = pg.NativeCircuit(ctx, dict(circuit, circuit_digest=None), compile_gates=compile_gates)
        got = nc.prove_bytes(wires, pis, salts=salts)
        nc.close()
        ok = got == exp
        print(f"case {k:3d} {'ok  ' if ok else 'FAIL'} {family:9s} 2^{degree_bits:<2d} {len(got):7d} B compile={int(compile_gates)} blinded={int(salts is not None)} cap={cap_height} arity={arity}", flush=True)
        if not ok:
            print(json.dumps(dict(failed_case=k, seed=seed, family=family, degree_bits=degree_bits, compile_gates=compile_gates, blinded=salts is not None)))
            sys.exit(1)
    print(json.dumps(dict(cases=cases, seed=seed, all_equal=True, oracle="oracle/prove_oracle.c", seconds=round(time.time() - t0, 1))))


if __name__ == "__main__":
    main()
