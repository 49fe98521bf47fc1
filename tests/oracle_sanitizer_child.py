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
print("asan/ubsan run ok")
