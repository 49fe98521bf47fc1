"""CPU model of the matrix-core Poseidon of csrc/poseidon.h, step by step as the kernel does it — byte planes of the state as
signed bytes (byte - 128), the 12 x 12 product per plane with the offset 128 * (row sum) in the accumulator, the 16-bit packing of
the plane sums, the two column sums with the NEXT round's constant as (X, Y) in their high dwords, the diagonal entry aside, one
reduction — against the textbook permutation of oracle/pyref.py and the reference's known answers. It reads the same generated
tables the kernel is compiled with (csrc/poseidon_limb_constants.h), so a table that no longer matches the constants, a plane sum
that leaves 16 bits or a column sum that could overflow fails here, without a GPU."""
import os
import re
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyref  # noqa: E402

P = pyref.P
HDR = open(os.path.join(ROOT, "plonky2_gpu_amd", "csrc", "poseidon_limb_constants.h")).read()


def table_u32(name):
    m = re.search(r"uint32_t %s\[(\d+)\][^=]*= \{(.*?)\};" % name, HDR, re.S)
    v = [int(t[:-1], 16) for t in re.findall(r"0x[0-9a-fA-F]+u", m.group(2))]
    assert len(v) == int(m.group(1))
    return v


XY = table_u32("POSEIDON_MDS_XY")
ROW_WORDS = [int(t[:-1], 16) for t in re.findall(r"0x[0-9a-fA-F]+u", re.search(r"#define POSEIDON_MDS_ROW_WORDS \{(.*?)\}", HDR).group(1))]
PLANE_OFFSET = int(re.search(r"#define POSEIDON_MDS_PLANE_OFFSET (\d+)", HDR).group(1))
DIAG0 = int(re.search(r"#define POSEIDON_MDS_DIAG0 (\d+)", HDR).group(1))


def matrix_row(r):
    """Row r of the A operand's matrix: words (4w + 12 - r) mod 12 of POSEIDON_MDS_ROW_WORDS, bytes in k order (mds_operands)."""
    out = []
    for w in range(3):
        word = ROW_WORDS[(4 * w + 12 - r) % 12]
        out += [(word >> (8 * t)) & 0xFF for t in range(4)]
    return out


def fold96(al, ah):
    assert al < 1 << 64 and (ah >> 32) < (1 << 32) - 1  # what gl::fold96 needs: the 64-bit sums exist, ah's high dword + carry fits
    return (al + (ah << 32)) % P


def mds_layer_model(s, layer):
    """One MDS layer + the constants of the round that follows, as csrc/poseidon.h mds_layer computes them."""
    rows = [matrix_row(r) for r in range(12)]
    planes = []
    for b in range(8):
        flipped = [((x >> (8 * b)) & 0xFF) ^ 0x80 for x in s]       # the byte the kernel hands to the matrix cores
        signed = [v - 256 if v >= 128 else v for v in flipped]       # ... which read it as a signed byte: byte - 128
        d = [sum(rows[r][j] * signed[j] for j in range(12)) + PLANE_OFFSET for r in range(12)]
        assert all(0 <= v < 1 << 16 for v in d), "a plane sum left its 16 bits"
        planes.append(d)
    out = []
    for r in range(12):
        a_l = planes[0][r] | planes[2][r] << 16
        b_l = planes[1][r] | planes[3][r] << 16
        a_h = planes[4][r] | planes[6][r] << 16
        b_h = planes[5][r] | planes[7][r] << 16
        x, y = XY[24 * layer + 2 * r], XY[24 * layer + 2 * r + 1]
        al = (x << 32 | a_l) + 256 * b_l
        ah = (y << 32 | a_h) + 256 * b_h
        if r == 0:
            al += DIAG0 * (s[0] & 0xFFFFFFFF)
            ah += DIAG0 * (s[0] >> 32)
        out.append(fold96(al, ah))
    return out


def permute_model(state):
    s = [(x + pyref.ROUND_CONSTANTS[i]) % P for i, x in enumerate(state)]
    for r in range(30):
        if r < 4 or r >= 26:
            s = [pow(x, 7, P) for x in s]
        else:
            s[0] = pow(s[0], 7, P)
        s = mds_layer_model(s, r)
    return s


def test_matrix_rows_are_the_mds_matrix():
    assert pyref.MDS_DIAG[0] == DIAG0 and not any(pyref.MDS_DIAG[1:])
    for r in range(12):
        assert matrix_row(r) == [pyref.MDS_CIRC[(j - r) % 12] for j in range(12)]
    assert PLANE_OFFSET == 128 * sum(pyref.MDS_CIRC) and sum(pyref.MDS_CIRC) * 255 < 1 << 16


def test_xy_pairs_solve_for_the_round_constants():
    for layer in range(30):
        for r in range(12):
            x, y = XY[24 * layer + 2 * r], XY[24 * layer + 2 * r + 1]
            c = pyref.ROUND_CONSTANTS[12 * (layer + 1) + r] if layer < 29 else 0
            assert (x << 32) + (y << 64) - c == 0 or ((x << 32) + (y << 64) - c) % P == 0
            assert x < (1 << 32) - (1 << 10) and y < (1 << 32) - (1 << 11)  # room for the 41-bit sums and the reduction's carry


def test_layer_and_permutation_equal_the_textbook():
    rng = np.random.default_rng(23)
    edge = [0, 1, P - 1, P, 2**64 - 1, 0x8080808080808080, 0x7F7F7F7F7F7F7F7F, 0xFF00FF00FF00FF00, 0xFFFFFFFF00000000, 0x00000000FFFFFFFF]
    states = [[edge[(i + 3 * j) % len(edge)] for j in range(12)] for i in range(len(edge))] + [[e] * 12 for e in edge]
    states += [[int(v) for v in rng.integers(0, 2**64, size=12, dtype=np.uint64)] for _ in range(40)]
    for s in states:
        for layer in (0, 3, 4, 25, 26, 29):  # any 64-bit representatives go in, as out of gl::pow7 / fold96
            exp = pyref._mds([x % P for x in s])
            nxt = pyref.ROUND_CONSTANTS[12 * (layer + 1):12 * (layer + 2)] if layer < 29 else [0] * 12
            assert mds_layer_model(s, layer) == [(e + c) % P for e, c in zip(exp, nxt)]
    for s in states[:12] + states[-12:]:
        assert permute_model([x % P for x in s]) == pyref.poseidon(s)
    # poseidon_goldilocks.rs:286-309 (first two vectors: zeros, 0..11)
    assert permute_model([0] * 12)[0] == 0x3C18A9786CB0B359
    assert permute_model(list(range(12)))[0] == 0xD64E1E3EFC5B8E9E
