"""accel.py — lets the proof-level restatements (prove_ref / fri_ref / plonk_ref, pure Python big ints) reach
2^10..2^14 rows: inside `with c_backend():` the four primitives they spend 97 % of their time in — the Poseidon
permutation and sponge, the Merkle tree, and the NTT — are served by the C restatement (oracle/gl_oracle.c through
oracle.py), which is pinned by the reference's own known answers (tests/test_oracle_*.py) and by the independent model in
pyref.py. Everything else (permutation argument, gates, quotient, challenger logic, FRI folding, openings, verifier)
stays the Python restatement. TEST INFRASTRUCTURE ONLY.

tests/test_oracle_prove.py::test_c_backend_gives_the_same_proof proves the same circuits with and without the backend
and demands identical proofs, so the accelerated oracle is the same oracle, only faster.
"""
import contextlib

import numpy as np

from . import oracle as o
from . import pyref

P = pyref.P


def _ints(a):
    return [int(x) for x in np.asarray(a).reshape(-1)]


def _poseidon(state):
    return _ints(o.canon(o.poseidon([x % P for x in state])))


def _hash_no_pad(inputs):
    return _ints(o.canon(o.hash_no_pad([x % P for x in inputs]))) if len(inputs) else pyref_hash_no_pad(inputs)


def _two_to_one(l, r):
    return _ints(o.canon(o.two_to_one([x % P for x in l], [x % P for x in r])))


def _fast_ntt(a, inverse=False):
    v = np.array([x % P for x in a], dtype=np.uint64)
    if len(a) == 1:
        return [int(v[0])]
    return _ints(o.canon(o.ifft(v) if inverse else o.fft(v)))


def _merkle_tree(leaves, cap_height):
    lens = {len(l) for l in leaves}
    assert len(lens) == 1
    lv = np.array([[x % P for x in l] for l in leaves], dtype=np.uint64)
    dig, cap = o.merkle_tree(lv, cap_height, threads=4)
    dig, cap = o.canon(dig), o.canon(cap)
    return [_ints(d) for d in dig], [_ints(c) for c in cap]


pyref_hash_no_pad = pyref.hash_no_pad


@contextlib.contextmanager
def c_backend():
    saved = {k: getattr(pyref, k) for k in ("poseidon", "hash_no_pad", "two_to_one", "fast_ntt", "merkle_tree")}
    pyref.poseidon, pyref.hash_no_pad, pyref.two_to_one = _poseidon, _hash_no_pad, _two_to_one
    pyref.fast_ntt, pyref.merkle_tree = _fast_ntt, _merkle_tree
    try:
        yield
    finally:
        for k, v in saved.items():
            setattr(pyref, k, v)
