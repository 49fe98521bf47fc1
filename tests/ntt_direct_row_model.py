# Index / twiddle / bank-conflict model of ntt_row_natural_direct_kernel (ntt_direct.hip)
import random, sys
P = 0xFFFFFFFF00000001
def root(lg): return pow(1753635133440165772, 1 << (32 - lg), P)
def brev(x, bits):
    r = 0
    for i in range(bits): r |= ((x >> i) & 1) << (bits - 1 - i)
    return r
def dif(v, w):
    n = len(v); lg = n.bit_length() - 1
    return [sum(v[i] * pow(w, i * brev(s, lg), P) for i in range(n)) % P for s in range(n)]
SQ, SB, SA = 136, 544, 8736
def conflicts_read(addrs):   # ds_read_b64: groups of 32 lanes, 64 banks of 4 bytes, each lane 2 banks
    worst = 1
    for g0 in (0, 32):
        cnt = {}
        for a in set(addrs[g0:g0 + 32]):
            for d in (0, 4):
                bk = ((a + d) // 4) % 64
                cnt[bk] = cnt.get(bk, 0) + 1
        worst = max(worst, max(cnt.values()))
    return worst
def conflicts_write(addrs):  # ds_write_b64: groups of 16 lanes, 32 banks
    worst = 1
    for g0 in range(0, 64, 16):
        cnt = {}
        for a in set(addrs[g0:g0 + 16]):
            for d in (0, 4):
                bk = ((a + d) // 4) % 32
                cnt[bk] = cnt.get(bk, 0) + 1
        worst = max(worst, max(cnt.values()))
    return worst
def run(inverse, seed=3):
    R = 1024; N1 = 64; log_n = 16; n = 1 << log_n; b = 2; row_shift = 1 if inverse else 0
    rng = random.Random(seed)
    rows = {}
    wR = root(10); w16 = pow(wR, 64, P); w64 = pow(wR, 16, P); w4 = pow(wR, 256, P)
    X = {}
    worst_r = worst_w = 1
    for w in range(16):
        row = (b * 16 + w + row_shift) & (N1 - 1)
        x = [rng.randrange(P) for _ in range(R)]
        rows[row] = x
        priv = {}
        # round 1 + private write
        A = {}
        for lane in range(64):
            v = dif([x[64 * i + lane] for i in range(16)], w16)
            for s in range(16):
                ka = brev(s, 4)
                v[s] = v[s] * pow(wR, ka * lane, P) % P
            A[lane] = v
        for s in range(16):
            ka = brev(s, 4)
            addrs = []
            for lane in range(64):
                q, h = lane & 3, lane >> 2
                a = h * SB + q * SQ + w * 8 + ka * SA
                assert a not in priv or True
                priv[a] = A[lane][s]; addrs.append(a)
            worst_w = max(worst_w, conflicts_write(addrs))
        assert len(priv) == 1024
        # private read, round 2, X write
        for h in range(16):
            addrs = [ (lane >> 2) * SA + (lane & 3) * SQ + w * 8 + h * SB for lane in range(64)]
            worst_r = max(worst_r, conflicts_read(addrs))
        A2 = {}
        for lane in range(64):
            q, ka = lane & 3, lane >> 2
            v = [priv[ka * SA + q * SQ + w * 8 + h * SB] for h in range(16)]
            v = dif(v, w16)
            A2[lane] = v
        for s in range(16):
            kb = brev(s, 4)
            addrs = []
            for lane in range(64):
                q, ka = lane & 3, lane >> 2
                val = A2[lane][s]   # the twiddle w_64^(kb q) is applied by the reader (as shifts: w_64 = 2^39)
                a = ka * SA + q * SQ + w * 8 + kb * SB
                X[a] = val; addrs.append(a)
            worst_w = max(worst_w, conflicts_write(addrs))
    assert len(X) == 16 * 1024
    out = {}
    assert w64 == pow(2, 39, P)
    for wv in range(16):   # wave = (kBhi, kAlo), lane = (kAhi, row)
        kbhi, kalo = wv >> 2, wv & 3
        for s in range(16):
            addrs = [(4 * (l >> 4) + kalo) * SA + kbhi * 4 * SB + (l & 15) * 8 + (s >> 2) * SB + (s & 3) * SQ for l in range(64)]
            worst_r = max(worst_r, conflicts_read(addrs))
        for lane in range(64):
            r, ka = lane & 15, 4 * (lane >> 4) + kalo
            B = [X[ka * SA + kbhi * 4 * SB + r * 8 + (s >> 2) * SB + (s & 3) * SQ] for s in range(16)]
            for s in range(16):   # a shift by (39 q kB) mod 96 and a sign (2^96 = -1)
                K = (39 * (s & 3) * (4 * kbhi + (s >> 2))) % 192
                B[s] = B[s] * pow(2, K % 96, P) * (P - 1 if K >= 96 else 1) % P
            for j in range(4):
                B[4 * j:4 * j + 4] = dif(B[4 * j:4 * j + 4], w4)
            k1 = (b * 16 + r + row_shift) & (N1 - 1)
            o_lane = k1 + N1 * (ka + 64 * kbhi)
            for s in range(16):
                kblo, kc = s >> 2, brev(s & 3, 2)
                o = o_lane + N1 * (16 * kblo + 256 * kc)
                if inverse: o = (-o) & (n - 1)
                assert o not in out
                out[o] = B[s]
    bad = 0
    for row, x in list(rows.items())[:6]:
        for k2 in random.Random(5).sample(range(R), 12):
            e = sum(x[j] * pow(wR, j * k2, P) for j in range(R)) % P
            o = row + N1 * k2
            if inverse: o = (n - o) % n
            if out.get(o) != e: bad += 1
    print("inverse", inverse, "mismatches", bad, "worst read conflict", worst_r, "worst write conflict", worst_w)
    return bad
sys.exit(1 if run(False) + run(True) else 0)
