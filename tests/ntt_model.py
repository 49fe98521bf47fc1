#!/usr/bin/env python3
"""Executable model of the index arithmetic of csrc/ntt.hip (pass kernel + planner).

Not part of the product: a CPU-side check of the four-step decomposition, digit order, twiddle
exponents, LDS slot permutation and store addressing, run on small sizes against the O(n log n)
definition in oracle/pyref.py. `python tests/ntt_model.py` exits non-zero on a mismatch.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pyref  # noqa: E402

P = pyref.P
LOGE = 13
E = 1 << LOGE
NT = 512


def brev(x, bits):
    return pyref.reverse_bits(x, bits) if bits else 0


def w(log, e):
    return pow(pyref.root_of_unity(log), e, P)


def radix_dif(v, base, D):
    for s in range(D - 1, -1, -1):
        half = 1 << s
        for bf in range((1 << D) // 2):
            blk, j = divmod(bf, half)
            i0 = base + blk * 2 * half + j
            i1 = i0 + half
            K = (39 * j * (32 >> s)) % 192
            a, c = v[i0], v[i1]
            v[i0] = (a + c) % P
            v[i1] = (a - c) * pow(2, K, P) % P


def run_pass(src, dst, LOGR, p, grid):
    """p: dict mirroring PassParams."""
    R = 1 << LOGR
    logt = p["logt"]
    T = 1 << logt
    fl = p["flags"]
    load_rows, store_rows, natural, inverse, coset = (fl & 1, fl & 2, fl & 4, fl & 8, fl & 16)
    D0 = LOGR % 4
    rounds = []
    if D0:
        rounds.append((D0, LOGR - D0))
    sh = LOGR - D0 - 4
    while sh >= 0:
        rounds.append((4, sh))
        sh -= 4
    for z in range(grid[2]):
        for a in range(grid[1]):
            for b in range(grid[0]):
                data = {}
                in_base = a * p["in_sa"] + b * p["in_sb"] + z * p.get("in_sz", 0)
                for m in range(R):
                    for t in range(T):
                        if b * T + t < p["t_limit"]:
                            o = in_base + (t * p["in_t"] + m if load_rows else t + m * p["in_m"])
                            val = src[o]
                            if coset and not load_rows:
                                val = val * pow(p["s"][z], m * p["in_m"], P) % P
                        else:
                            val = 0
                        data[(m, t)] = val
                for (D, SH) in rounds:
                    RD, G = 1 << D, 16 >> D
                    new = {}
                    for tid in range(NT):
                        for g in range(G):
                            gid = tid + g * NT
                            l, rest = gid & (T - 1), gid >> logt
                            rest_lo, rest_hi = rest & ((1 << SH) - 1), rest >> SH
                            mbase = (rest_hi << (SH + D)) | rest_lo
                            v = [data[(mbase | (i << SH), l)] for i in range(RD)]
                            radix_dif(v, 0, D)
                            if SH > 0:
                                for i in range(1, RD):
                                    k1 = brev(i, D)
                                    e = (rest_lo * k1) << (LOGR - SH - D)
                                    v[i] = v[i] * w(LOGR, e) % P
                            elif p.get("twiddle"):
                                assert D == 4 and G == 1
                                L = b * T + l
                                kr = brev(rest, LOGR - 4)
                                c = w(p["tw_hi"], L * kr)
                                if coset:
                                    c = c * pow(p["s"][z], L, P) % P
                                step = w(p["tw_hi"], L << (LOGR - 4))
                                for j in range(16):
                                    i = brev(j, 4)
                                    v[i] = v[i] * c % P
                                    c = c * step % P
                            nat = natural and SH == 0
                            kr2 = brev(rest_hi, LOGR - D) if nat else 0
                            for i in range(RD):
                                m = ((brev(i, D) << (LOGR - D)) | kr2) if nat else (mbase | (i << SH))
                                new[(m, l)] = v[i]
                    assert len(new) == len(data)
                    data = new
                zo = brev(z, p.get("rate_bits", 0)) if coset else z
                out_base = a * p["out_sa"] + b * p["out_sb"] + zo * p.get("out_sz", 0)
                n = 1 << p["log_n"]
                for m in range(R):
                    for t in range(T):
                        if b * T + t >= p["t_limit"]:
                            continue
                        val = data[(m, t)] * p.get("scale", 1) % P
                        o = out_base + (t * p["out_t"] + m if store_rows else t + m * p["out_m"])
                        if inverse:
                            o = (o & ~(n - 1)) | ((n - (o & (n - 1))) & (n - 1))
                        dst[o] = val


def ntt_batch(src, n_polys, log_n, natural, inverse):
    n = 1 << log_n
    dst = list(src)
    n_inv = pow(n, P - 2, P) if inverse else 1
    fnat, finv = (4 if natural else 0), (8 if inverse else 0)
    if log_n <= 12:
        logt = LOGE - log_n
        T = 1 << logt
        p = dict(logt=logt, t_limit=n_polys, in_sa=0, in_sb=T * n, in_t=n, in_m=1, out_sa=0, out_sb=T * n, out_t=n,
                 out_m=1, flags=1 | 2 | fnat | finv, log_n=log_n, scale=n_inv)
        run_pass(src, dst, log_n, p, ((n_polys + T - 1) // T, 1, 1))
        return dst
    la = (log_n + 1) // 2
    lb = log_n - la
    N1, N2 = 1 << la, 1 << lb
    logtA = LOGE - la
    TA = 1 << logtA
    p = dict(logt=logtA, t_limit=N2, in_sa=n, in_sb=TA, in_t=1, in_m=N2, out_sa=n, out_sb=TA, out_t=1, out_m=N2,
             flags=fnat, log_n=log_n, tw_hi=log_n, twiddle=True)
    run_pass(src, dst, la, p, (N2 // TA, n_polys, 1))
    logtB = LOGE - lb
    TB = 1 << logtB
    p = dict(logt=logtB, t_limit=N1, in_sa=n, in_sb=TB * N2, in_t=N2, in_m=1, out_sa=n, log_n=log_n, scale=n_inv)
    if natural:
        p.update(out_sb=TB, out_t=1, out_m=N1, flags=1 | 4 | finv)
    else:
        p.update(out_sb=TB * N2, out_t=N2, out_m=1, flags=1 | 2)
    src2 = list(dst)
    run_pass(src2, dst, lb, p, (N1 // TB, n_polys, 1))
    return dst


def coset_lde(coeffs, n_polys, log_n, rate_bits, shift=7):
    """two-pass fused path only (log_n >= 13)."""
    n = 1 << log_n
    nc = 1 << rate_bits
    s = [shift * w(log_n + rate_bits, r) % P for r in range(nc)]
    dst = [0] * (n_polys * n * nc)
    la = (log_n + 1) // 2
    lb = log_n - la
    N1, N2 = 1 << la, 1 << lb
    logtA = LOGE - la
    TA = 1 << logtA
    p = dict(logt=logtA, t_limit=N2, in_sa=n, in_sb=TA, in_sz=0, in_t=1, in_m=N2, out_sa=n * nc, out_sb=TA, out_sz=n,
             out_t=1, out_m=N2, flags=16, log_n=log_n, tw_hi=log_n, twiddle=True, s=s, rate_bits=rate_bits)
    run_pass(coeffs, dst, la, p, (N2 // TA, n_polys, nc))
    logtB = LOGE - lb
    TB = 1 << logtB
    p = dict(logt=logtB, t_limit=N1, in_sa=n * nc, in_sb=TB * N2, in_sz=n, in_t=N2, in_m=1, out_sa=n * nc,
             out_sb=TB * N2, out_sz=n, out_t=N2, out_m=1, flags=1 | 2, log_n=log_n)
    src2 = list(dst)
    run_pass(src2, dst, lb, p, (N1 // TB, n_polys, nc))
    return dst


def main():
    g = pyref.splitmix64(1)
    ok = True
    for log_n, n_polys in [(1, 3), (2, 2), (3, 2), (4, 2), (5, 1), (6, 1), (7, 1), (9, 1), (10, 1), (12, 1), (13, 1)]:
        n = 1 << log_n
        x = [next(g) for _ in range(n * n_polys)]
        exp_f = sum((pyref.fast_ntt(x[i * n:(i + 1) * n]) for i in range(n_polys)), [])
        exp_i = sum((pyref.fast_ntt(x[i * n:(i + 1) * n], inverse=True) for i in range(n_polys)), [])
        got = ntt_batch(x, n_polys, log_n, True, False)
        r1 = got == exp_f
        got = ntt_batch(x, n_polys, log_n, True, True)
        r2 = got == exp_i
        got = ntt_batch(x, n_polys, log_n, False, False)
        exp_b = [exp_f[(i // n) * n + brev(i % n, log_n)] for i in range(n * n_polys)]
        r3 = got == exp_b
        print(f"log_n={log_n:2d} polys={n_polys} natural={r1} inverse={r2} bitrev={r3}")
        ok &= r1 and r2 and r3
    # fused coset LDE, smallest two-pass size
    log_n, rate = 13, 1
    n = 1 << log_n
    c = [next(g) for _ in range(n)]
    got = coset_lde(c, 1, log_n, rate)
    scaled = [x * pow(7, i, P) % P for i, x in enumerate(c)] + [0] * (n * ((1 << rate) - 1))
    nat = pyref.fast_ntt(scaled)
    exp = [nat[brev(i, log_n + rate)] for i in range(n << rate)]
    r = got == exp
    print(f"coset_lde log_n={log_n} rate={rate}: {r}")
    ok &= r
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
