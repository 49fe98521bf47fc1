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


# ---- the blocked partial rounds (the EXPERIMENT tools/experiments/poseidon_blocked.h partial_block, tables of tools/gen_poseidon_block_tables.py) -------------------------
BHDR = open(os.path.join(ROOT, "tools", "experiments", "poseidon_block_constants.h")).read()


def block_table(name):
    m = re.search(r"uint32_t %s\[(\d+)\][^=]*= \{(.*?)\};" % name, BHDR, re.S)
    v = [int(t[:-1], 16) if t.startswith("0x") else int(t[:-1]) for t in re.findall(r"(?:0x[0-9a-fA-F]+|\d+)u", m.group(2))]
    assert len(v) == int(m.group(1))
    return v


def block_define(name):
    return int(re.search(r"#define %s (\d+)" % name, BHDR).group(1))


BT, BPLANES, BBIAS, BCOUNT, BFIRST, BROWS = (block_define("POSEIDON_BLOCK_" + n) for n in ("T", "PLANES", "BIAS", "COUNT", "FIRST_ROUND", "ROWS"))
BLOCK_A, BLOCK_UX, BLOCK_H = (block_table("POSEIDON_BLOCK_" + n) for n in ("A", "U", "H"))
assert all(BLOCK_UX[64 * b:64 * b + 64] == BLOCK_UX[:64] for b in range(BCOUNT))
BLOCK_U, BLOCK_XC = BLOCK_UX[:48], BLOCK_UX[48:64]


def s8(b):
    return b - 256 if b >= 128 else b


def block_matrix(block, q, p):
    """the sixteen signed bytes of logical row q, digit plane p, as the matrix cores read the A operand"""
    words = BLOCK_A[((block * 16 + q) * BT + p) * 4:((block * 16 + q) * BT + p) * 4 + 4]
    return [s8((w >> (8 * t)) & 0xFF) for w in words for t in range(4)]


def partial_block_model(s, block):
    """T partial rounds (constants of the block's first round already in s; the constants of the round after the block come out
    added), step by step as the kernel: biased plane chains on the matrix cores, pairs, group sums with their high-dword
    constants, al = G0 - G2, ah = G1 + G2, the x chain with its s-boxes, the rank-one terms, one fold per row."""
    # B operand: plane k = byte k of the twelve words as byte - 128, with the constant word 0x8080808080808081 in element 0's place
    # (the signed byte 1 in plane 0, zeros elsewhere: N has no column 0, the rows' spare constants ride there); slots 12-15 hold
    # whatever the registers hold (here: an arbitrary pattern) and meet zero columns of A
    fed = [0x8080808080808081] + list(s[1:])
    b_op = [[s8(((x >> (8 * k)) & 0xFF) ^ 0x80) for x in fed] + [s8((37 * k + 11 * j) & 0xFF) for j in range(4)] for k in range(8)]
    d = []
    for w in range(BPLANES):
        acc = [BBIAS] * 16
        for p in range(BT):
            k = w - p
            if 0 <= k < 8:
                for q in range(16):
                    acc[q] += sum(a * b for a, b in zip(block_matrix(block, q, p), b_op[k]))
        assert BBIAS == 0 and all(-(1 << 20) < v < 1 << 20 for v in acc), "a plane sum left (-2^20, 2^20)"
        d.append(acc)
    al, ah = [], []
    for q in range(BROWS):
        e = [d[2 * j][q] + (d[2 * j + 1][q] << 8) if 2 * j + 1 < BPLANES else d[2 * j][q] for j in range((BPLANES + 1) // 2)]
        assert len(e) == 6 and all(-(1 << 31) <= v < 1 << 31 for v in e)  # 32-bit signed arithmetic
        hl, hh = BLOCK_H[block * 32 + q], BLOCK_H[block * 32 + 16 + q]
        g0 = (hl << 32) + e[0] + (e[1] << 16)   # 64-bit signed multiply-add chains
        g1 = (hh << 32) + e[2] + (e[3] << 16)
        g2 = e[4] + (e[5] << 16)
        # the kernel's 64-bit arithmetic wraps; what must hold is that the TRUE sums al, ah are non-negative 64-bit numbers
        assert -(1 << 63) <= g2 < 1 << 63 and 0 <= g0 - g2 < 1 << 64 and 0 <= g1 + g2 < 1 << 64
        al.append(g0 - g2)
        ah.append(g1 + g2)
    # the x chain: x_0 = s[0]; x_t from row 11 + t and the earlier y's
    y = []
    x = s[0]
    for t in range(BT):
        if t:
            a_l = al[11 + t] + sum(BLOCK_XC[t * BT + u] * (y[u] & 0xFFFFFFFF) for u in range(t))
            a_h = ah[11 + t] + sum(BLOCK_XC[t * BT + u] * (y[u] >> 32) for u in range(t))
            x = fold96(a_l, a_h)
        y.append(pow(x, 7, P) + (P if (x * 7 + t) % 3 == 0 and pow(x, 7, P) < (1 << 64) - P else 0))  # any representative, as gl::pow7 may return
    out = []
    for q in range(12):
        a_l = al[q] + sum(BLOCK_U[(BT - 1 - t) * 12 + q] * (y[t] & 0xFFFFFFFF) for t in range(BT))
        a_h = ah[q] + sum(BLOCK_U[(BT - 1 - t) * 12 + q] * (y[t] >> 32) for t in range(BT))
        out.append(fold96(a_l, a_h))
    return out


def permute_model_blocked(state):
    s = [(x + pyref.ROUND_CONSTANTS[i]) % P for i, x in enumerate(state)]
    r = 0
    while r < 30:
        if r < 4 or r >= 26:
            s = mds_layer_model([pow(x, 7, P) for x in s], r)
            r += 1
        elif r >= BFIRST and (r - BFIRST) % BT == 0 and (r - BFIRST) // BT < BCOUNT:
            s = partial_block_model(s, (r - BFIRST) // BT)
            r += BT
        else:
            s = mds_layer_model([pow(s[0], 7, P)] + s[1:], r)
            r += 1
    return s


def test_block_tables_are_current_and_cover_the_partial_rounds():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_poseidon_block_tables as g

    assert g.generate() == BHDR, "re-run tools/gen_poseidon_block_tables.py"
    assert BFIRST + BT * BCOUNT == 26 and BFIRST >= 4 and BPLANES == 8 + BT - 1 and BROWS == 12 + BT - 1
    # the matrix digits recombine to N^T and the rows of the lower powers; spare column 12 carries the row's constant only
    m, powers, m0 = g.matrices(open(g.SRC).read())
    rows = g.logical_rows(powers)
    for block in range(BCOUNT):
        for q in range(16):
            for j in range(1, 12):
                assert sum(block_matrix(block, q, p)[j] << (8 * p) for p in range(BT)) == rows[q][j]
            assert all(block_matrix(block, q, p)[12:] == [0, 0, 0, 0] for p in range(BT))
            assert rows[q][0] == 0  # column 0 of every power of N is zero: the spare constant's place
    # worst-case bounds, re-derived: every plane chain has at most T terms of at most 12 * 128 * 128 + 128
    assert BT * 12 * 128 * 128 < 1 << 20
    for block in range(BCOUNT):
        for q in range(BROWS):
            usum = sum(BLOCK_U[j * 12 + q] for j in range(BT)) if q < 12 else sum(BLOCK_XC[(q - 11) * BT + u] for u in range(BT))
            g0, g1, g2, ymax = g.group_bounds(usum)
            hl, hh = BLOCK_H[block * 32 + q], BLOCK_H[block * 32 + 16 + q]
            assert (hl << 32) - g0 - g2 >= 0 and (hl << 32) + g0 + g2 + ymax < 1 << 64
            assert (hh << 32) - g1 - g2 >= 0 and (((hh << 32) + g1 + g2 + ymax) >> 32) + 1 < (1 << 32) - 1


def test_blocked_partial_rounds_equal_the_textbook():
    rng = np.random.default_rng(29)
    edge = [0, 1, P - 1, P, 2**64 - 1, 0x8080808080808080, 0x7F7F7F7F7F7F7F7F, 0xFF00FF00FF00FF00, 0xFFFFFFFF00000000, 0x00000000FFFFFFFF]
    states = [[edge[(i + 3 * j) % len(edge)] for j in range(12)] for i in range(len(edge))] + [[e] * 12 for e in edge]
    states += [[int(v) for v in rng.integers(0, 2**64, size=12, dtype=np.uint64)] for _ in range(30)]
    for s in states:
        for block in range(BCOUNT):
            r0 = BFIRST + BT * block
            exp = [x % P for x in s]
            for r in range(r0, r0 + BT):  # T textbook partial rounds; round r's constants are in, round r+1's are added after the layer
                exp[0] = pow(exp[0], 7, P)
                exp = pyref._mds(exp)
                nxt = pyref.ROUND_CONSTANTS[12 * (r + 1):12 * (r + 2)]
                exp = [(e + c) % P for e, c in zip(exp, nxt)]
            assert partial_block_model(s, block) == exp, (block, s)
    for s in states[:8] + states[-8:]:
        assert permute_model_blocked([x % P for x in s]) == pyref.poseidon(s)
    assert permute_model_blocked([0] * 12)[0] == 0x3C18A9786CB0B359
    assert permute_model_blocked(list(range(12)))[0] == 0xD64E1E3EFC5B8E9E
