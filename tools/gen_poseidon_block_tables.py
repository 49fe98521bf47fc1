#!/usr/bin/env python3
"""Tables for the BLOCKED partial rounds of the matrix-core Poseidon (tools/experiments/poseidon_blocked.h partial_block): T = 4 partial rounds of
`poseidon_naive` (plonky2/src/hash/poseidon.rs:565-585) as ONE pass over the matrix cores.

In a partial round only element 0 goes through the s-box; with M the MDS matrix, Z = diag(0, 1, .., 1), N = M Z, m0 = M e_0:
    v_{t+1} = N v_t + y_t m0 + c_{r+t+1},      y_t = (v_t[0])^7
so over a block of T rounds that starts from v_0 (round r's constants already added)
    v_T = N^T v_0 + sum_t y_t N^(T-1-t) m0 + sum_{j=1..T} N^(T-j) c_{r+j}
    x_t = v_t[0] = (N^t v_0)[0] + sum_{u<t} y_u (N^(t-1-u) m0)[0] + (sum_{j<=t} N^(t-j) c_{r+j})[0].
The linear parts of v_0 — twelve rows of N^T and row 0 of N, N^2, .., N^(T-1): fifteen of the sixteen result rows a lane of
v_mfma_i32_32x32x32_i8 receives — are integer products of the state's byte planes with the matrix entries' SIGNED base-256
digits (entries of N^4 have 29 bits: four digits in [-128, 127]); products of equal weight 256^(p+k) are chained in the
accumulator, so the vector ALU packs and reduces ONCE per block instead of once per round. The y_t terms are rank-one
corrections with small integer weights (below 2^29), two multiply-adds per row and round.

Written to tools/experiments/poseidon_block_constants.h (an EXPERIMENT, see tools/experiments/poseidon_blocked.h: correct, fewer instructions, not faster):
  POSEIDON_BLOCK_A[blocks][16][T][4]  the A operand: logical row q (0-11 rows of N^T, 11 + t = row 0 of N^t, 15 unused), digit plane p,
                               bytes of columns 0-3 | 4-7 | 8-11 | 12-15 as two's-complement bytes; column 0 (zero in every power of N) holds the
                               digits of the row's spare constant e (the kernel feeds the signed byte 1 there in plane 0), columns 12-15 zero
  POSEIDON_BLOCK_U[blocks][64] u_j = N^j m0 (weights of y_{T-1-j} in the new state), [T][12], then xc[t][u] = (N^(t-1-u) m0)[0] (weight of
                               y_u in x_t, u < t), [T][T]; the same in every block
  POSEIDON_BLOCK_H[blocks][2][16]  per block: Hl of the rows, Hh of the rows (padded to 16): the high dwords of the group sums G0, G1, chosen with e such that
                               Hl 2^32 + Hh 2^64 + e = (additive constant of the row) + (offset of the signed bytes) - (accumulator bias)
                               (mod p) and every sum stays inside 64 bits (tests/test_poseidon_matrix_model.py re-derives the bounds)
`--check` exits 1 if the committed file is stale."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "plonky2_gpu_amd", "csrc", "poseidon_constants.h")
OUT = os.path.join(ROOT, "tools", "experiments", "poseidon_block_constants.h")
P = 0xFFFFFFFF00000001
W = 12
T = 4
HALF_FULL, N_PARTIAL = 4, 22
BLOCK_ROUNDS = [4 + 2 + T * b for b in range((N_PARTIAL - 2) // T)]  # rounds 4, 5 stay plain layers; blocks start at 6, 10, 14, 18, 22
PLANES = 8 + T - 1     # weights 256^0 .. 256^(7 + T - 1)
BIAS = 0               # the chains start from zero: plane sums are SIGNED, |chain| < 2^20 (a 2^20 bias would cost sixteen registers)
ROWS = 12 + T - 1      # logical rows in use


def table(text, name):
    m = re.search(r"uint64_t %s\[(\d+)\] = \{(.*?)\};" % name, text, re.S)
    vals = [int(t[:-3], 16) for t in re.findall(r"0x[0-9a-fA-F]+ULL", m.group(2))]
    assert len(vals) == int(m.group(1))
    return vals


def matmul(a, b):
    return [[sum(a[i][k] * b[k][j] for k in range(W)) for j in range(W)] for i in range(W)]


def matvec(a, v):
    return [sum(a[i][k] * v[k] for k in range(W)) for i in range(W)]


def matrices(text):
    circ, diag = table(text, "POSEIDON_MDS_CIRC"), table(text, "POSEIDON_MDS_DIAG")
    m = [[0] * W for _ in range(W)]
    for r in range(W):  # row r: sum_i circ[i] s[(i + r) % 12] + diag[r] s[r]   (poseidon.rs:174-194)
        for i in range(W):
            m[r][(i + r) % W] += circ[i]
        m[r][r] += diag[r]
    n = [[0 if j == 0 else m[i][j] for j in range(W)] for i in range(W)]  # N = M Z
    powers = [[[int(i == j) for j in range(W)] for i in range(W)]]
    for _ in range(T):
        powers.append(matmul(powers[-1], n))
    m0 = [m[i][0] for i in range(W)]
    return m, powers, m0


def digits(e):
    """signed base-256 digits of e >= 0, T of them, each in [-128, 127]"""
    out = []
    for _ in range(T):
        d = ((e + 128) % 256) - 128
        out.append(d)
        e = (e - d) // 256
    assert e == 0, "entry needs more than T digits"
    return out


def logical_rows(powers):
    """rows 0-11: N^T; row 11 + t: row 0 of N^t (t = 1 .. T-1); the rest zero"""
    rows = [list(r) for r in powers[T]] + [list(powers[t][0]) for t in range(1, T)]
    return rows + [[0] * W] * (16 - len(rows))


def group_bounds(u_sum):
    """worst-case magnitudes of the three (signed) group sums, without their high-dword constants, and the maximum of the rank-one
    terms of a row (non-negative)"""
    dmax = (1 << 20) - 1  # |chain| <= T * (11 * 128 * 128 + 128 * 128) < 2^20
    g0 = dmax * (1 + (1 << 8) + (1 << 16) + (1 << 24))
    g1 = g0
    g2 = dmax * sum(1 << (8 * i) for i in range(PLANES - 8))
    y = ((1 << 32) - 1) * u_sum  # sum_t (32-bit half of y_t) * weight
    return g0, g1, g2, y


def signed_digits(e):
    """T signed base-256 digits in [-128, 127] of a (possibly negative) integer"""
    out = []
    for _ in range(T):
        d = ((e + 128) % 256) - 128
        out.append(d)
        e = (e - d) // 256
    assert e == 0, "constant needs more than T digits"
    return out


def solve_h(c, u_sum):
    """(Hl, Hh, e) with Hl 2^32 + Hh 2^64 + e == c (mod p): Hl, Hh are the high dwords of the group sums G0, G1 (G2's is zero) and
    e, a signed 32-bit integer, is the row's entry in the matrix's SPARE column 0 (N has a zero column 0: element 0 enters the block
    only through the s-box; the kernel feeds the constant word 0x8080808080808081 in its place, i.e. the signed byte 1 in plane 0
    and zeros elsewhere), i.e. a free additive constant of weight 1. e is what makes every constant solvable: it moves the low
    dword of the target, which fixes Hh, to wherever the range conditions want it. For every possible state:
         al = G0 - G2 + rank-one terms  in [0, 2^64)                      with G0 = g0 + Hl 2^32
         ah = G1 + G2 + rank-one terms  with high dword + 1 < 2^32 - 1 (gl::fold96)"""
    g0, g1, g2, y = group_bounds(u_sum)
    c %= P
    c_lo = c & 0xFFFFFFFF
    lim = (1 << 31) - (1 << 25)
    for step in range(0, 1 << 32, 1 << 25):
        for k in (0, 1):
            hh = step
            low = (-(hh + k)) % (1 << 32)      # the low dword the shifted target must have
            e = c_lo - low
            if not -lim <= e <= lim:
                continue
            t = (c - e) % P
            if (-(t + k)) % (1 << 32) != hh:
                continue
            num = t + k * P + hh
            assert num % (1 << 32) == 0
            hl = (num >> 32) - hh
            if not 0 <= hl < 1 << 32:
                continue
            assert ((hl << 32) + (hh << 64) + e - c) % P == 0
            al_min = (hl << 32) - g0 - g2
            al_max = (hl << 32) + g0 + g2 + y
            ah_min = (hh << 32) - g1 - g2
            ah_max = (hh << 32) + g1 + g2 + y
            if al_min >= 0 and al_max < 1 << 64 and ah_min >= 0 and (ah_max >> 32) + 1 < (1 << 32) - 1:
                return hl, hh, e
    raise SystemExit("no (Hl, Hh, e) for constant 0x%016x" % c)


def build():
    text = open(SRC).read()
    rc = table(text, "POSEIDON_ALL_ROUND_CONSTANTS")
    m, powers, m0 = matrices(text)
    rows = logical_rows(powers)
    assert max(max(r) for r in powers[T]) < 1 << 29
    dg_rows = [[digits(e) for e in rows[q]] for q in range(16)]  # [16][12 columns][T]
    u = [matvec(powers[j], m0) for j in range(T)]  # u_j = N^j m0
    assert max(max(v) for v in u) < 1 << 29
    xc = [[u[t - 1 - uu][0] if uu < t else 0 for uu in range(T)] for t in range(T)]
    ones = ((1 << 64) - 1) // 255  # sum_k 256^k, k < 8
    bias_total = BIAS * sum(1 << (8 * w) for w in range(PLANES))
    h_tab, a_tabs = [], []
    for r0 in BLOCK_ROUNDS:
        def consts(t):  # sum_{j=1..t} N^(t-j) c_{r0+j}
            acc = [0] * W
            for j in range(1, t + 1):
                c = rc[12 * (r0 + j):12 * (r0 + j + 1)] if r0 + j < 30 else [0] * W
                acc = [(a + b) % P for a, b in zip(acc, matvec(powers[t - j], c))]
            return acc
        kvec = consts(T)
        block, spare = [], []
        for q in range(ROWS):
            if q < 12:
                add, usum = kvec[q], sum(u[j][q] for j in range(T))
            else:
                t = q - 11
                add, usum = consts(t)[0], sum(xc[t])
            off = 128 * ones * sum(rows[q])
            hl, hh, e = solve_h((add + off - bias_total) % P, usum)
            block.append((hl, hh))
            spare.append(signed_digits(e))
        spare += [[0] * T] * (16 - ROWS)
        h_tab.append(block)
        # the A operand of THIS block: [16][T][4 dwords]; column 0 (zero in N^t) = digit p of e_q; dword 3 = 0
        for q in range(16):
            assert all(dg_rows[q][0][p] == 0 for p in range(T)), "column 0 of N^t is not zero"
        a_tabs.append([[[sum(((spare[q][p] if (w, t) == (0, 0) else dg_rows[q][4 * w + t][p]) & 0xFF) << (8 * t) for t in range(4)) for w in range(3)] + [0]
                        for p in range(T)] for q in range(16)])
    return a_tabs, u, xc, h_tab


def generate():
    a_tabs, u, xc, h_tab = build()
    out = ["/* Poseidon: tables of the blocked partial rounds on the matrix cores (tools/experiments/poseidon_blocked.h partial_block; derivation in",
           " * tools/gen_poseidon_block_tables.py, which generates this file from poseidon_constants.h). Do not edit by hand. */",
           "#pragma once", "#include <stdint.h>", "#ifndef POSEIDON_CONST", "#define POSEIDON_CONST static const", "#endif",
           f"#define POSEIDON_BLOCK_T {T}", f"#define POSEIDON_BLOCK_PLANES {PLANES}", f"#define POSEIDON_BLOCK_BIAS {BIAS}",
           f"#define POSEIDON_BLOCK_COUNT {len(BLOCK_ROUNDS)}", f"#define POSEIDON_BLOCK_FIRST_ROUND {BLOCK_ROUNDS[0]}",
           f"#define POSEIDON_BLOCK_ROWS {ROWS}", ""]
    out.append("/* [block][logical row q][digit plane p][dword]: bytes of matrix columns 0-3 | 4-7 | 8-11 | 12-15 (column 0 = the row's spare constant, 12-15 zero) */")
    out.append(f"POSEIDON_CONST uint32_t POSEIDON_BLOCK_A[{len(a_tabs) * 16 * T * 4}] __attribute__((aligned(64))) = {{")
    for a_tab in a_tabs:
        for q in range(16):
            out.append("    " + ", ".join(f"0x{w:08x}u" for p in range(T) for w in a_tab[q][p]) + ",")
    out.append("};")
    out.append("/* [block][64]: [j][row] u_j = N^j m0, the weight of y_(T-1-j) in the new state (48), then [t][u] the weight of y_u in x_t (16).")
    out.append(" * The same for every block; one copy per block so that the kernel's (uniform) loads depend on the block and the values live in")
    out.append(" * scalar registers only while a block runs instead of through the whole kernel. */")
    out.append(f"POSEIDON_CONST uint32_t POSEIDON_BLOCK_U[{len(a_tabs) * 64}] __attribute__((aligned(64))) = {{")
    for _ in a_tabs:
        for j in range(T):
            out.append("    " + ", ".join(f"{v}u" for v in u[j]) + ",")
        out.append("    " + ", ".join(f"{v}u" for t in range(T) for v in xc[t]) + ",")
    out.append("};")
    out.append("/* [block][32]: Hl of the fifteen rows, one pad, Hh of the fifteen rows, one pad — the high dwords of the group sums G0, G1; two")
    out.append(" * 64-byte scalar loads per block */")
    out.append(f"POSEIDON_CONST uint32_t POSEIDON_BLOCK_H[{len(h_tab) * 32}] __attribute__((aligned(64))) = {{")
    for block in h_tab:
        for half in (0, 1):
            out.append("    " + ", ".join(f"0x{hs[half]:08x}u" for hs in block) + ", 0u,")
    out.append("};")
    out.append("")
    return "\n".join(out)


if __name__ == "__main__":
    src = generate()
    if "--check" in sys.argv:
        sys.exit(0 if os.path.exists(OUT) and open(OUT).read() == src else 1)
    open(OUT, "w").write(src)
    print("wrote", OUT)
