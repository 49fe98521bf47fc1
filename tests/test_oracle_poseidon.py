"""Pins the C oracle's Poseidon / sponge / Merkle tree with the reference's known answers
(plonky2/src/hash/poseidon_goldilocks.rs:277-318, plonky2/src/hash/merkle_tree.rs:443-515)."""
import numpy as np
import pytest

from oracle import pyref

P = pyref.P
NEG1 = P - 1

# plonky2/src/hash/poseidon_goldilocks.rs:286-309 (expected outputs from a modified hadeshash reference)
TEST_VECTORS = [
    ([0] * 12,
     [0x3c18a9786cb0b359, 0xc4055e3364a246c3, 0x7953db0ab48808f4, 0xc71603f33a1144ca,
      0xd7709673896996dc, 0x46a84e87642f44ed, 0xd032648251ee0b3c, 0x1c687363b207df62,
      0xdf8565563e8045fe, 0x40f5b37ff4254dae, 0xd070f637b431067c, 0x1792b1c4342109d7]),
    (list(range(12)),
     [0xd64e1e3efc5b8e9e, 0x53666633020aaa47, 0xd40285597c6a8825, 0x613a4f81e81231d2,
      0x414754bfebd051f0, 0xcb1f8980294a023f, 0x6eb2a9e4d54a9d0f, 0x1902bc3af467e056,
      0xf045d5eafdc6021f, 0xe4150f77caaa3be5, 0xc9bfd01d39b50cce, 0x5c0a27fcb0e1459b]),
    ([NEG1] * 12,
     [0xbe0085cfc57a8357, 0xd95af71847d05c09, 0xcf55a13d33c1c953, 0x95803a74f4530e82,
      0xfcd99eb30a135df1, 0xe095905e913a3029, 0xde0392461b42919b, 0x7d3260e24e81d031,
      0x10d3d0465d9deaa0, 0xa87571083dfc2a47, 0xe18263681e9958f8, 0xe28e96f1ae5e60d3]),
    ([0x8ccbbbea4fe5d2b7, 0xc2af59ee9ec49970, 0x90f7e1a9e658446a, 0xdcc0630a3ab8b1b8,
      0x7ff8256bca20588c, 0x5d99a7ca0c44ecfb, 0x48452b17a70fbee3, 0xeb09d654690b6c88,
      0x4a55d3a39c676a88, 0xc0407a38d2285139, 0xa234bac9356386d1, 0xe1633f2bad98a52f],
     [0xa89280105650c4ec, 0xab542d53860d12ed, 0x5704148e9ccab94f, 0xd3a826d4b62da9f5,
      0x8a7a6ca87892574f, 0xc7017e1cad1a674e, 0x1f06668922318e34, 0xa3b203bc8102676f,
      0xfcc781b0ce382bf2, 0x934c69ff3ed14ba5, 0x504688a5996e8f13, 0x401f3f2ed524a2ba]),
]


@pytest.mark.parametrize("inp,exp", TEST_VECTORS)
def test_poseidon_known_answers(oracle, inp, exp):
    assert oracle.canon(oracle.poseidon(inp)).tolist() == exp
    assert oracle.canon(oracle.poseidon(inp, naive=True)).tolist() == exp
    assert pyref.poseidon(inp) == exp


def test_fast_equals_naive_partial_rounds(oracle):
    # poseidon.rs:736-749 (input 0..11) + random and non-canonical inputs
    assert (oracle.canon(oracle.poseidon(range(12))) == oracle.canon(oracle.poseidon(range(12), naive=True))).all()
    for seed in range(20):
        x = oracle.random_field(12, seed=seed)
        assert (oracle.canon(oracle.poseidon(x)) == oracle.canon(oracle.poseidon(x, naive=True))).all()
    x = np.array([2**64 - 1] * 12, dtype=np.uint64)  # non-canonical representative of 2^32-2
    assert oracle.canon(oracle.poseidon(x)).tolist() == pyref.poseidon([2**64 - 1] * 12)


def test_sponge_lengths(oracle):
    for ln in [0, 1, 3, 4, 5, 7, 8, 9, 15, 16, 17, 20, 135]:
        x = oracle.random_field(ln, seed=100 + ln)
        assert oracle.canon(oracle.hash_or_noop(x)).tolist() == pyref.hash_or_noop(x.tolist())
        if ln > 0:
            assert oracle.canon(oracle.hash_no_pad(x)).tolist() == pyref.hash_no_pad(x.tolist())
    l, r = oracle.random_field(4, seed=1), oracle.random_field(4, seed=2)
    assert oracle.canon(oracle.two_to_one(l, r)).tolist() == pyref.two_to_one(l.tolist(), r.tolist())
    # hash_or_noop canonicalises short inputs (config.rs:56-67)
    assert oracle.hash_or_noop(np.array([P + 3, 5], dtype=np.uint64)).tolist() == [3, 5, 0, 0]


@pytest.mark.parametrize("cap_height", [1, 8, 0, 4])
def test_merkle_every_leaf_verifies(oracle, cap_height):
    # merkle_tree.rs:456-514: 256 random leaves of 7 elements; cap_height 1 and log_n (=8)
    n, k = 256, 7
    leaves = oracle.random_field((n, k), seed=cap_height)
    dig, cap = oracle.merkle_tree(leaves, cap_height, threads=2)
    for i in range(n):
        sib = oracle.merkle_prove(dig, n, cap_height, i)
        assert sib.shape[0] == 8 - cap_height
        assert oracle.merkle_verify(leaves[i], i, cap, sib)
    # a corrupted leaf must not verify
    bad = leaves[3].copy()
    bad[0] ^= np.uint64(1)
    assert not oracle.merkle_verify(bad, 3, cap, oracle.merkle_prove(dig, n, cap_height, 3))


def test_merkle_cap_height_too_big(oracle):
    # merkle_tree.rs:470-482 (should_panic)
    with pytest.raises(ValueError):
        oracle.merkle_tree(oracle.random_field((256, 7)), 9)


@pytest.mark.parametrize("n,k,h", [(16, 7, 0), (16, 7, 2), (32, 3, 1), (8, 135, 3), (64, 9, 4), (4, 4, 2)])
def test_merkle_layout_matches_independent_model(oracle, n, k, h):
    leaves = oracle.random_field((n, k), seed=n * 1000 + k * 10 + h)
    dig, cap = oracle.merkle_tree(leaves, h)
    pd, pc = pyref.merkle_tree(leaves.tolist(), h)
    assert oracle.canon(dig).tolist() == pd
    assert oracle.canon(cap).tolist() == pc


def test_commit_from_values_matches_independent_model(oracle):
    P_, lg, rate, h = 5, 4, 3, 2
    vals = oracle.random_field((P_, 1 << lg), seed=4242)
    got = oracle.commit_from_values(vals, rate, h, threads=2)
    coeffs, leaves, dig, cap = pyref.commit_from_values(vals.tolist(), rate, h)
    assert oracle.canon(got["coeffs"]).tolist() == coeffs
    assert oracle.canon(got["leaves"]).tolist() == leaves
    assert oracle.canon(got["digests"]).tolist() == dig
    assert oracle.canon(got["cap"]).tolist() == cap


def test_device_limb_tables_are_what_the_generator_writes_and_what_they_claim():
    """csrc/poseidon_limb_constants.h (c, c*2^21, c*2^42 mod p in the order the device consumes them) is generated data: it
    must equal the generator's output, and every triple must satisfy x*c = x0*c + x1*c21 + x2*c42 for a 21/21/22-bit split."""
    import os
    import random
    import re
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    assert subprocess.run([sys.executable, os.path.join(root, "tools", "gen_poseidon_limb_tables.py"), "--check"]).returncode == 0
    text = open(os.path.join(root, "plonky2_gpu_amd", "csrc", "poseidon_limb_constants.h")).read()
    P = 0xFFFFFFFF00000001

    def arr(name):
        m = re.search(r"uint64_t %s\[(\d+)\][^=]*= \{(.*?)\};" % name, text, re.S)
        v = [int(t[:-3], 16) for t in re.findall(r"0x[0-9a-fA-F]+ULL", m.group(2))]
        assert len(v) == int(m.group(1))
        return v

    rng = random.Random(3)
    for stream, n in (("POSEIDON_INIT_STREAM", 121), ("POSEIDON_PARTIAL_STREAM", 594)):
        c0, c21, c42 = arr(stream + "_C0"), arr(stream + "_C21"), arr(stream + "_C42")
        assert len(c0) == n
        for k in range(n):
            assert c21[k] == c0[k] * (1 << 21) % P and c42[k] == c0[k] * (1 << 42) % P
            x = rng.randrange(1 << 64)
            x0, x1, x2 = x & 0x1FFFFF, (x >> 21) & 0x1FFFFF, x >> 42
            lo = x0 * (c0[k] & 0xFFFFFFFF) + x1 * (c21[k] & 0xFFFFFFFF) + x2 * (c42[k] & 0xFFFFFFFF)
            hi = x0 * (c0[k] >> 32) + x1 * (c21[k] >> 32) + x2 * (c42[k] >> 32)
            assert lo < 1 << 56 and hi < 1 << 56  # 23 terms stay far below 2^63 (gl::fold96's domain)
            assert (lo + (hi << 32)) % P == x * c0[k] % P
