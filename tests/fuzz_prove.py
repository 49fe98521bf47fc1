"""Differential campaign: randomly shaped small circuits through gl_prove (native prover), each proof
compared byte for byte with the oracle's (oracle/prove_ref.py). Shapes vary in degree, selector
grouping, FRI arities, rate, cap height, proof-of-work bits, query count, quotient degree factor,
number of challenges and gate compilation. Not part of the test suite; run it on a GPU box:
    python tests/fuzz_prove.py [cases=40] [seed=1]
Prints one line per case and a JSON summary; exits non-zero on the first mismatch."""
import json
import os
import random
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import plonky2_gpu_amd as pg  # noqa: E402
from oracle import prove_ref, serialize_ref  # noqa: E402
from plonk_instance import make_circuit  # noqa: E402


def random_shape(rng):
    degree_bits = rng.choice([3, 4, 4, 5, 5, 6])
    two_groups = rng.random() < 0.5
    rate_bits = rng.choice([1, 2, 3, 3])
    qdf_min = 4 if two_groups else 5
    qdf = rng.choice([q for q in (4, 5, 6, 7, 8) if qdf_min <= q <= (1 << rate_bits)] or [None])
    if qdf is None:  # the quotient degree must fit the rate (prover.rs:807-811)
        rate_bits, qdf = 3, rng.choice([q for q in (5, 6, 7, 8) if q >= qdf_min])
    # reduction arities: keep every layer at least as large as the cap (reduction_strategies.rs:38-48)
    cap_height = rng.choice([0, 1, 2])
    arity, bits = [], degree_bits
    while bits > 0 and rng.random() < 0.75:
        ab = rng.choice([1, 1, 2, 3, 4])
        if ab > bits or bits + rate_bits - ab < cap_height:
            break
        arity.append(ab)
        bits -= ab
    return dict(degree_bits=degree_bits, two_groups=two_groups, rate_bits=rate_bits, cap_height=cap_height, arity_bits=tuple(arity),
                pow_bits=rng.choice([0, 1, 3, 6]), num_queries=rng.choice([1, 2, 4]), quotient_degree_factor=qdf,
                num_challenges=rng.choice([1, 2, 2, 3]))


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = random.Random(seed)
    ctx = pg.Context(0)
    t0 = time.time()
    for k in range(cases):
        shape = random_shape(rng)
        compile_gates = rng.random() < 0.5
        circuit, wires, pis = make_circuit(seed=1000 * seed + k, **shape)
        exp = serialize_ref.proof_bytes(prove_ref.prove(circuit, wires, pis))
        nc = pg.NativeCircuit(ctx, dict(circuit, circuit_digest=None), compile_gates=compile_gates)
        got = nc.prove_bytes(wires, pis)
        nc.close()
        ok = got == exp
        print(f"case {k:3d} {'ok  ' if ok else 'FAIL'} {len(got):6d} B compile={int(compile_gates)} {shape}", flush=True)
        if not ok:
            print(json.dumps(dict(failed_case=k, seed=seed, shape=shape, compile_gates=compile_gates)))
            sys.exit(1)
    print(json.dumps(dict(cases=cases, seed=seed, all_equal=True, seconds=round(time.time() - t0, 1))))


if __name__ == "__main__":
    main()
