# Index/twiddle model of ntt_col_direct_kernel (ntt_direct.hip): exact arithmetic, one tile.
import random, sys
P = 0xFFFFFFFF00000001
def root(lg): return pow(1753635133440165772, 1 << (32 - lg), P)
def brev(x, bits):
    r = 0
    for i in range(bits): r |= ((x >> i) & 1) << (bits - 1 - i)
    return r
def dif(v, w):  # slot s holds frequency brev(s); w = primitive len(v)-th root
    n = len(v); lg = n.bit_length() - 1
    out = [0] * n
    for s in range(n):
        k = brev(s, lg)
        out[s] = sum(v[i] * pow(w, i * k, P) for i in range(n)) % P
    return out
def run(LOGG, natural, log_n=20, b=5, seed=1):
    G = 1 << LOGG; LOGC = 6 - LOGG; C = 1 << LOGC; LOGR = 8 + LOGG; R = 1 << LOGR
    rng = random.Random(seed)
    x = [[rng.randrange(P) for c in range(C)] for m in range(R)]
    wR = root(LOGR); w16 = pow(wR, R // 16, P); wG = pow(wR, R // G, P) if G > 1 else 1; w16G = pow(wR, R // (16 * G), P); wn = root(log_n)
    # first rounds
    X = {}   # (kab, w, c) -> value
    for w in range(16):
        priv = {}
        for g in range(G):
            for c in range(C):
                A = [x[(R // 16) * i + 16 * g + w][c] for i in range(16)]
                A = dif(A, w16)
                for s in range(16):
                    ka = brev(s, 4)
                    A[s] = A[s] * pow(wR, ka * (16 * g + w), P) % P
                    priv[(16 * g + ka, c)] = A[s]       # private row 16 g + kA
        for g in range(G):      # reader lane (g, c)
            for c in range(C):
                n = [0] * 16
                for j in range(16 // G):
                    for gg in range(G):
                        n[j * G + gg] = priv[(16 * gg + g + G * j, c)]
                for j in range(16 // G):
                    if G > 1:
                        n[j * G:(j + 1) * G] = dif(n[j * G:(j + 1) * G], wG)
                    for s2 in range(G):
                        kb = brev(s2, LOGG)
                        val = n[j * G + s2] * pow(w16G, kb * w, P) % P
                        X[(g + G * j + 16 * kb, w, c)] = val
    # last rounds
    out = [[None] * C for _ in range(R)]
    for wv in range(16):
        for g in range(G):
            for c in range(C):
                kab = G * wv + g
                B = [X[(kab, wr, c)] for wr in range(16)]
                B = dif(B, w16)
                L = b * C + c
                cc = pow(wn, L * kab, P); step = pow(wn, L << (4 + LOGG), P)
                row_lane = kab if natural else (brev(kab & 15, 4) << (LOGR - 4)) | (brev(kab >> 4, LOGG) << 4)
                for j in range(16):
                    s3 = brev(j, 4)
                    val = B[s3] * cc % P
                    cc = cc * step % P
                    row = row_lane + ((j << (4 + LOGG)) if natural else s3)
                    assert out[row][c] is None
                    out[row][c] = val
    # expected
    bad = 0
    for c in range(0, C, max(1, C // 4)):
        L = b * C + c
        col = [x[m][c] for m in range(R)]
        for k1 in random.Random(7).sample(range(R), 24):
            e = sum(col[m] * pow(wR, m * k1, P) for m in range(R)) % P * pow(wn, L * k1, P) % P
            row = k1 if natural else brev(k1, LOGR)
            if out[row][c] != e: bad += 1
    print("LOGG", LOGG, "natural", natural, "mismatches", bad)
    return bad
tot = 0
for lg in (3, 2, 1, 0):
    for nat in (True, False):
        tot += run(lg, nat)
sys.exit(1 if tot else 0)
