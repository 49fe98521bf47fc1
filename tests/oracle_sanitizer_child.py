"""Child of tests/test_oracle_sanitizers.py: drives the C oracle (built with -fsanitize=address,undefined, path in argv[1]) through its
whole API on small shapes, including the ragged and degenerate ones. Any sanitizer report ends the process with an error."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle.oracle as o  # noqa: E402

o._LIB_PATH = sys.argv[1]
o.build = lambda force=False: o._LIB_PATH
rng = np.random.default_rng(1)
def rf(shape): return o.random_field(shape, seed=int(rng.integers(1, 1 << 30)))
for lg in (0, 1, 2, 3, 5, 8, 12, 13):
    x = rf((3, 1 << lg))
    f = o.fft_batch(x, threads=2); assert (o.canon(o.fft_batch(f, inverse=True, threads=2)) == x).all()
    for r in (0, 1, 3):
        o.coset_lde(x[0], r)
    o.coset_ifft(o.coset_fft(x[0], 7), 7); o.fft(x[0], r=min(lg, 2))
for n, k, h in ((1, 9, 0), (2, 5, 1), (16, 3, 0), (16, 4, 2), (64, 135, 3), (256, 7, 8), (8, 1, 1)):
    lv = rf((n, k)); dig, cap = o.merkle_tree(lv, h, threads=2)
    for i in (0, n - 1):
        sib = o.merkle_prove(dig, n, h, i); assert o.merkle_verify(lv[i], i, cap, sib)
for P_, lg, r, h in ((5, 4, 3, 2), (135, 6, 3, 4), (3, 0, 3, 1), (20, 10, 3, 4), (4, 5, 3, 8)):
    v = rf((P_, 1 << lg)); e = o.commit_from_values(v, r, h, threads=3); o.commit_from_coeffs(o.canon(e["coeffs"]), r, h, threads=3)
s = rf((12,)); assert (o.poseidon(s) == o.poseidon(s, naive=True)).all()
for ln in (0, 1, 4, 5, 8, 9, 135): o.hash_or_noop(rf((ln,))) if ln else None
o.root_table_concat(16); o.fft_bench(1 << 10, 2, 1)
# the prover above the commit (oracle/prove_oracle.c): whole proofs of the small circuits — one selector group and several, the circuit with
# every ed25519 gate kind, the recursion-shaped one with the eight upstream kinds, a blinded one, a failing quotient, every gate on a
# random row — through the same sanitized library
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from oracle import accel, gates_ref, prove_c  # noqa: E402
from plonk_instance import FULL_GATES, RECURSION_GATES, make_circuit, make_full_circuit, make_recursion_circuit  # noqa: E402

for kw in (dict(degree_bits=4, seed=3), dict(degree_bits=5, seed=4, two_groups=True, arity_bits=(3,)), dict(degree_bits=3, seed=12, arity_bits=()),
           dict(degree_bits=4, seed=9, quotient_degree_factor=5), dict(degree_bits=4, seed=11, num_challenges=3)):
    c, w, pis = make_circuit(**kw)
    assert len(prove_c.prove(c, w, pis, threads=2, trace={})) > 1000
with accel.c_backend():
    for mk in (make_full_circuit, make_recursion_circuit):
        c, w, pis = mk(4, seed=2)
        assert len(prove_c.prove(c, w, pis, threads=3)) > 1000
c, w, pis = make_circuit(4, seed=14)
c = dict(c, fri_params=dict(c["fri_params"], hiding=True))
prove_c.prove(c, w, pis, salts=o.random_field((3, 4, 1 << 7), seed=5), threads=2)
c, w, pis = make_circuit(4, seed=25, quotient_degree_factor=5)
w = [list(col) for col in w]
w[3] = [(v + 1) % o.P for v in w[3]]
try:
    prove_c.prove(c, w, pis, threads=2)
    raise SystemExit("a broken witness must fail")
except AssertionError as e:
    assert "Quotient has failed" in str(e)
import random  # noqa: E402
r2 = random.Random(3)
for kind, param in FULL_GATES + RECURSION_GATES:
    row = [r2.randrange(o.P) for _ in range(max(gates_ref.num_wires(kind, param), 1))]
    prove_c.gate_constraints(kind, param, [1, 2], row, [3, 4, 5, 6])
print("asan/ubsan run ok")
