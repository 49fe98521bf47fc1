# Index / twiddle / bank-conflict model of the in-place direct row pass (1024-point rows, bit-reversed output)
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
def conf_r(addrs):
    worst = 1
    for g0 in (0, 32):
        cnt = {}
        for a in set(addrs[g0:g0 + 32]):
            bk = (a // 8) % 32; cnt[bk] = cnt.get(bk, 0) + 1
        worst = max(worst, max(cnt.values()))
    return worst
def conf_w(addrs):
    worst = 1
    for g0 in range(0, 64, 16):
        cnt = {}
        for a in set(addrs[g0:g0 + 16]):
            bk = (a // 8) % 16; cnt[bk] = cnt.get(bk, 0) + 1
        worst = max(worst, max(cnt.values()))
    return worst
rng = random.Random(4)
x = [rng.randrange(P) for _ in range(1024)]
wR = root(10); w16 = pow(wR, 64, P); w64 = pow(wR, 16, P); w4 = pow(wR, 256, P)
ww = wr = 1
# R1 + E1
buf = {}
A = {}
for lane in range(64):
    v = dif([x[64 * i + lane] for i in range(16)], w16)
    for s in range(16):
        v[s] = v[s] * pow(wR, brev(s, 4) * lane, P) % P
    A[lane] = v
def e1_addr(ka, h, q): return (ka * 4 + h * 68 + q) * 8
for s in range(16):
    ka = brev(s, 4); addrs = []
    for lane in range(64):
        a = e1_addr(ka, lane >> 2, lane & 3); buf[a] = A[lane][s]; addrs.append(a)
    ww = max(ww, conf_w(addrs))
B = {}
for h in range(16):
    addrs = [e1_addr(l >> 2, h, l & 3) for l in range(64)]; wr = max(wr, conf_r(addrs))
for lane in range(64):
    ka, q = lane >> 2, lane & 3
    v = dif([buf[e1_addr(ka, h, q)] for h in range(16)], w16)
    for s in range(16):
        v[s] = v[s] * pow(w64, brev(s, 4) * q, P) % P
    B[lane] = v
# E2: slot = kB*64 + kA*4 + (q ^ (kB >> 2))
def e2_addr(ka, kb, q): return (kb * 64 + ka * 4 + (q ^ (kb >> 2))) * 8
buf = {}
for s in range(16):
    kb = brev(s, 4); addrs = []
    for lane in range(64):
        a = e2_addr(lane >> 2, kb, lane & 3); assert a not in buf; buf[a] = B[lane][s]; addrs.append(a)
    ww = max(ww, conf_w(addrs))
C = {}
for r in range(16):
    addrs = [e2_addr(l >> 2, 4 * (l & 3) + (r >> 2), r & 3) for l in range(64)]; wr = max(wr, conf_r(addrs))
for lane in range(64):
    ka, kbhi = lane >> 2, lane & 3
    v = [buf[e2_addr(ka, 4 * kbhi + (r >> 2), r & 3)] for r in range(16)]
    for j in range(4):
        v[4 * j:4 * j + 4] = dif(v[4 * j:4 * j + 4], w4)
    C[lane] = v
# E3: slot = (pos & ~3) | ((pos & 3) ^ (pos >> 8)); pos = brev4(kA)*64 + brev2(kBlo)*16 + brev2(kBhi)*4 + s2
def e3_addr(pos): return ((pos & ~3) | ((pos & 3) ^ (pos >> 8))) * 8
buf = {}
for r in range(16):
    kblo, s2 = r >> 2, r & 3; addrs = []
    for lane in range(64):
        ka, kbhi = lane >> 2, lane & 3
        pos = brev(ka, 4) * 64 + brev(kblo, 2) * 16 + brev(kbhi, 2) * 4 + s2
        a = e3_addr(pos); assert a not in buf; buf[a] = C[lane][r]; addrs.append(a)
        # check that the value is frequency k2 with brev10(k2) == pos
        k2 = ka + 16 * (4 * kbhi + kblo) + 256 * brev(s2, 2)
        assert brev(k2, 10) == pos
    ww = max(ww, conf_w(addrs))
out = [None] * 1024
for r in range(16):
    addrs = [e3_addr(r * 64 + l) for l in range(64)]; wr = max(wr, conf_r(addrs))
    for l in range(64):
        out[r * 64 + l] = buf[e3_addr(r * 64 + l)]
bad = 0
for k2 in random.Random(5).sample(range(1024), 40):
    e = sum(x[j] * pow(wR, j * k2, P) for j in range(1024)) % P
    if out[brev(k2, 10)] != e: bad += 1
print("mismatches", bad, "worst write conflict", ww, "worst read conflict", wr)
sys.exit(1 if bad or ww > 1 or wr > 1 else 0)
