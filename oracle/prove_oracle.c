/* prove_oracle.c — CPU oracle for the prover above the commit (TEST INFRASTRUCTURE ONLY; see prove_oracle.h for scope and
 * parity status). Every function cites the reference file:line it restates. Threading (OpenMP) follows the reference's
 * rayon split: per column (fri/oracle.rs:720, 990-997; plonk/proof.rs:314-320), per point batch (plonk/prover.rs:884),
 * per subgroup row (prover.rs:746), per leaf chunk (fri/prover.rs:91-95).
 */
#include "prove_oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "gl_oracle.h"
#include "poseidon_constants.h"

#define SALT_SIZE 4 /* fri/oracle.rs:41 */
#define COSET_SHIFT 7ULL /* F::coset_shift() = MULTIPLICATIVE_GROUP_GENERATOR, types.rs:431-433, goldilocks_field.rs:82 */
#define UNUSED_SELECTOR 0xFFFFFFFFULL /* gates/selectors.rs:11 */
#define SPONGE_RATE 8
#define SPONGE_WIDTH 12

static double now_s(void) {
#ifdef _OPENMP
    return omp_get_wtime();
#else
    return 0;
#endif
}

static unsigned log2_strict_sz(size_t n) {
    unsigned l = 0;
    while (((size_t)1 << l) < n) l++;
    return l;
}
static unsigned log2_ceil_u(uint32_t n) { /* util/src/lib.rs log2_ceil */
    unsigned l = 0;
    while (((uint32_t)1 << l) < n) l++;
    return l;
}

/* ------------------------------------------------------------------ F_p^2 = F_p[X]/(X^2 - 7) */
/* field/src/goldilocks_extensions.rs:19 (W = 7), field/src/extension/quadratic.rs:173-185 */
typedef struct {
    uint64_t a, b;
} e2;
static inline e2 e2_make(uint64_t a, uint64_t b) {
    e2 r = {a, b};
    return r;
}
static inline e2 e2_add(e2 x, e2 y) { return e2_make(glo_add(x.a, y.a), glo_add(x.b, y.b)); }
static inline e2 e2_sub(e2 x, e2 y) { return e2_make(glo_sub(x.a, y.a), glo_sub(x.b, y.b)); }
static inline e2 e2_mul(e2 x, e2 y) {
    uint64_t c0 = glo_add(glo_mul(x.a, y.a), glo_mul(7, glo_mul(x.b, y.b)));
    uint64_t c1 = glo_add(glo_mul(x.a, y.b), glo_mul(x.b, y.a));
    return e2_make(c0, c1);
}
static inline e2 e2_scalar(e2 x, uint64_t k) { return e2_make(glo_mul(x.a, k), glo_mul(x.b, k)); }
static inline int e2_is_one(e2 x) { return glo_canon(x.a) == 1 && glo_canon(x.b) == 0; }
static e2 e2_pow(e2 x, uint64_t e) {
    e2 acc = e2_make(1, 0);
    while (e) {
        if (e & 1) acc = e2_mul(acc, x);
        x = e2_mul(x, x);
        e >>= 1;
    }
    return acc;
}

/* ------------------------------------------------------------------ gates */
/* plonk_common.rs:116-128 reduce_with_powers: sum terms[i] * base^i */
static uint64_t reduce_with_powers_base(const uint64_t *terms, size_t n, uint64_t base) {
    uint64_t acc = 0;
    for (size_t i = n; i-- > 0;) acc = glo_add(glo_mul(acc, base), terms[i]);
    return acc;
}
/* prod_{k < bound} (x - k) */
static uint64_t range_product(uint64_t x, uint64_t bound) {
    uint64_t acc = 1;
    for (uint64_t k = 0; k < bound; k++) acc = glo_mul(acc, glo_sub(x, k));
    return acc;
}

/* --- Poseidon gate layers over field elements (hash/poseidon.rs *_field variants used by gates/poseidon.rs) */
static void pg_constant_layer(uint64_t *s, int rc) {
    for (int i = 0; i < 12; i++) s[i] = glo_add(s[i], POSEIDON_ALL_ROUND_CONSTANTS[rc * 12 + i]);
}
static uint64_t pg_sbox(uint64_t x) {
    uint64_t x2 = glo_mul(x, x), x4 = glo_mul(x2, x2);
    return glo_mul(glo_mul(x, x2), x4);
}
static void pg_sbox_mds(uint64_t *s) { /* sbox_layer_field + mds_layer_field (poseidon.rs:262-275, 531-540) */
    uint64_t t[12], out[12];
    for (int i = 0; i < 12; i++) t[i] = pg_sbox(s[i]);
    for (int r = 0; r < 12; r++) {
        uint64_t acc = glo_mul(t[r], POSEIDON_MDS_DIAG[r]);
        for (int i = 0; i < 12; i++) acc = glo_add(acc, glo_mul(t[(i + r) % 12], POSEIDON_MDS_CIRC[i]));
        out[r] = acc;
    }
    memcpy(s, out, sizeof out);
}
static void pg_partial_init(uint64_t *s) { /* partial_first_constant_layer + mds_partial_layer_init (poseidon.rs:312-365) */
    uint64_t t[12], out[12];
    for (int i = 0; i < 12; i++) t[i] = glo_add(s[i], POSEIDON_FAST_PARTIAL_FIRST_ROUND_CONSTANT[i]);
    out[0] = t[0];
    for (int c = 1; c < 12; c++) {
        uint64_t acc = 0;
        for (int r = 1; r < 12; r++) acc = glo_add(acc, glo_mul(t[r], POSEIDON_FAST_PARTIAL_ROUND_INITIAL_MATRIX[(r - 1) * 11 + (c - 1)]));
        out[c] = acc;
    }
    memcpy(s, out, sizeof out);
}
static void pg_partial_fast(uint64_t *s, int r) { /* mds_partial_layer_fast_field (poseidon.rs:429-450) */
    uint64_t out[12];
    uint64_t d = glo_mul(s[0], POSEIDON_MDS_CIRC[0] + POSEIDON_MDS_DIAG[0]);
    for (int i = 1; i < 12; i++) d = glo_add(d, glo_mul(s[i], POSEIDON_FAST_PARTIAL_ROUND_W_HATS[r * 11 + i - 1]));
    out[0] = d;
    for (int i = 1; i < 12; i++) out[i] = glo_add(s[i], glo_mul(s[0], POSEIDON_FAST_PARTIAL_ROUND_VS[r * 11 + i - 1]));
    memcpy(s, out, sizeof out);
}

/* gates/poseidon.rs:485-564 (eval_unfiltered_base_one); wire layout :40-105 */
static int poseidon_gate(const uint64_t *w, uint64_t *out) {
    enum { WIRE_SWAP = 24, START_DELTA = 25, START_FULL_0 = 29, START_PARTIAL = 29 + 36, START_FULL_1 = 29 + 36 + 22 };
    int k = 0;
    uint64_t swap = w[WIRE_SWAP];
    out[k++] = glo_mul(swap, glo_sub(swap, 1));
    for (int i = 0; i < 4; i++) out[k++] = glo_sub(glo_mul(swap, glo_sub(w[i + 4], w[i])), w[START_DELTA + i]);
    uint64_t s[12];
    for (int i = 0; i < 4; i++) {
        s[i] = glo_add(w[i], w[START_DELTA + i]);
        s[i + 4] = glo_sub(w[i + 4], w[START_DELTA + i]);
    }
    for (int i = 8; i < 12; i++) s[i] = w[i];
    int rc = 0;
    for (int r = 0; r < 4; r++) {
        pg_constant_layer(s, rc);
        if (r != 0)
            for (int i = 0; i < 12; i++) {
                uint64_t sin = w[START_FULL_0 + 12 * (r - 1) + i];
                out[k++] = glo_sub(s[i], sin);
                s[i] = sin;
            }
        pg_sbox_mds(s);
        rc++;
    }
    pg_partial_init(s);
    for (int r = 0; r < 22; r++) {
        uint64_t sin = w[START_PARTIAL + r];
        out[k++] = glo_sub(s[0], sin);
        s[0] = pg_sbox(sin);
        if (r < 21) s[0] = glo_add(s[0], POSEIDON_FAST_PARTIAL_ROUND_CONSTANTS[r]);
        pg_partial_fast(s, r);
    }
    rc += 22;
    for (int r = 0; r < 4; r++) {
        pg_constant_layer(s, rc);
        for (int i = 0; i < 12; i++) {
            uint64_t sin = w[START_FULL_1 + 12 * r + i];
            out[k++] = glo_sub(s[i], sin);
            s[i] = sin;
        }
        pg_sbox_mds(s);
        rc++;
    }
    for (int i = 0; i < 12; i++) out[k++] = glo_sub(s[i], w[12 + i]);
    return k;
}

static inline e2 wext(const uint64_t *w, size_t at) { return e2_make(w[at], w[at + 1]); } /* vars.get_local_ext, plonk/vars.rs:122-129 */

/* the barycentric interpolation the two interpolation gates constrain; defined below */
static int interpolation_gate(int low_degree, unsigned subgroup_bits, const uint64_t *w, uint64_t *out);

int glo_gate_num_constraints(const glo_gate *g) {
    const uint32_t *p = g->params;
    switch (g->kind) {
    case GLO_GATE_NOOP: return 0;
    case GLO_GATE_CONSTANT: return (int)p[0];
    case GLO_GATE_PUBLIC_INPUT: return 4;
    case GLO_GATE_ARITHMETIC: return (int)p[0];
    case GLO_GATE_BASE_SUM: return 1 + (int)p[1];
    case GLO_GATE_U32_ADD_MANY: return (int)p[1] * (3 + 18);
    case GLO_GATE_U32_ARITHMETIC: return (int)p[0] * (4 + 32);
    case GLO_GATE_U32_SUBTRACTION: return (int)p[0] * (3 + 16);
    case GLO_GATE_U32_RANGE_CHECK: return (int)p[0] * 17;
    case GLO_GATE_COMPARISON: return 6 + 5 * (int)p[1] + (int)((p[0] + p[1] - 1) / p[1]);
    case GLO_GATE_RANDOM_ACCESS: return (int)p[1] * ((int)p[0] + 2) + (int)p[2];
    case GLO_GATE_POSEIDON: return 12 * 7 + 22 + 12 + 1 + 4;
    case GLO_GATE_ARITHMETIC_EXTENSION: return 2 * (int)p[0];        /* arithmetic_extension.rs num_constraints = num_ops * D */
    case GLO_GATE_MUL_EXTENSION: return 2 * (int)p[0];               /* multiplication_extension.rs */
    case GLO_GATE_REDUCING: return 2 * (int)p[0];                    /* reducing.rs: D * num_coeffs */
    case GLO_GATE_REDUCING_EXTENSION: return 2 * (int)p[0];          /* reducing_extension.rs */
    case GLO_GATE_EXPONENTIATION: return (int)p[0] + 1;              /* exponentiation.rs: num_power_bits + 1 */
    case GLO_GATE_POSEIDON_MDS: return 12 * 2;                       /* poseidon_mds.rs: SPONGE_WIDTH * D */
    case GLO_GATE_LOW_DEGREE_INTERPOLATION: return (1 << p[0]) * 2 + 2 + 3 * ((1 << p[0]) - 2); /* low_degree_interpolation.rs:505-510 */
    case GLO_GATE_HIGH_DEGREE_INTERPOLATION: return 2 * (1 << p[0]) + 2; /* high_degree_interpolation.rs:207-211 */
    default: return -1;
    }
}

int glo_gate_constraints(const glo_gate *g, const uint64_t *consts, const uint64_t *w, const uint64_t pih[4], uint64_t *out) {
    const uint32_t *p = g->params;
    int k = 0;
    switch (g->kind) {
    case GLO_GATE_NOOP: /* gates/noop.rs */
        return 0;
    case GLO_GATE_CONSTANT: /* gates/constant.rs:150-158 */
        for (uint32_t i = 0; i < p[0]; i++) out[k++] = glo_sub(consts[i], w[i]);
        return k;
    case GLO_GATE_PUBLIC_INPUT: /* gates/public_input.rs:129-139 */
        for (int i = 0; i < 4; i++) out[k++] = glo_sub(w[i], pih[i]);
        return k;
    case GLO_GATE_ARITHMETIC: /* gates/arithmetic_base.rs:199-216 */
        for (uint32_t i = 0; i < p[0]; i++) {
            uint64_t computed = glo_add(glo_mul(glo_mul(w[4 * i], w[4 * i + 1]), consts[0]), glo_mul(w[4 * i + 2], consts[1]));
            out[k++] = glo_sub(w[4 * i + 3], computed);
        }
        return k;
    case GLO_GATE_BASE_SUM: { /* gates/base_sum.rs:213-230 */
        uint32_t B = p[0], nl = p[1];
        out[k++] = glo_sub(reduce_with_powers_base(w + 1, nl, B), w[0]);
        for (uint32_t i = 0; i < nl; i++) out[k++] = range_product(w[1 + i], B);
        return k;
    }
    case GLO_GATE_U32_ADD_MANY: { /* u32/src/gates/add_many_u32.rs:143-184 */
        uint32_t na = p[0], ops = p[1];
        for (uint32_t i = 0; i < ops; i++) {
            size_t o = (size_t)(na + 3) * i;
            uint64_t computed = w[o + na]; /* carry */
            for (uint32_t j = 0; j < na; j++) computed = glo_add(computed, w[o + j]);
            uint64_t res = w[o + na + 1], car = w[o + na + 2];
            out[k++] = glo_sub(glo_add(glo_mul(car, 1ULL << 32), res), computed);
            uint64_t comb_res = 0, comb_car = 0;
            for (int j = 17; j >= 0; j--) {
                uint64_t limb = w[(size_t)(na + 3) * ops + 18 * i + (size_t)j];
                out[k++] = range_product(limb, 4);
                if (j < 16)
                    comb_res = glo_add(glo_mul(4, comb_res), limb);
                else
                    comb_car = glo_add(glo_mul(4, comb_car), limb);
            }
            out[k++] = glo_sub(comb_res, res);
            out[k++] = glo_sub(comb_car, car);
        }
        return k;
    }
    case GLO_GATE_U32_ARITHMETIC: { /* u32/src/gates/arithmetic_u32.rs:326-385 */
        uint32_t ops = p[0];
        for (uint32_t i = 0; i < ops; i++) {
            const uint64_t *r = w + 6 * (size_t)i;
            uint64_t m0 = r[0], m1 = r[1], ad = r[2], lo = r[3], hi = r[4], inv = r[5];
            uint64_t computed = glo_add(glo_mul(m0, m1), ad);
            uint64_t diff = glo_sub(0xFFFFFFFFULL, hi);
            uint64_t hi_not_max = glo_sub(glo_mul(inv, diff), 1);
            out[k++] = glo_mul(hi_not_max, lo);
            out[k++] = glo_sub(glo_add(glo_mul(hi, 1ULL << 32), lo), computed);
            uint64_t c_lo = 0, c_hi = 0;
            for (int j = 31; j >= 0; j--) {
                uint64_t limb = w[6 * (size_t)ops + 32 * (size_t)i + (size_t)j];
                out[k++] = range_product(limb, 4);
                if (j < 16)
                    c_lo = glo_add(glo_mul(c_lo, 4), limb);
                else
                    c_hi = glo_add(glo_mul(c_hi, 4), limb);
            }
            out[k++] = glo_sub(c_lo, lo);
            out[k++] = glo_sub(c_hi, hi);
        }
        return k;
    }
    case GLO_GATE_U32_SUBTRACTION: { /* u32/src/gates/subtraction_u32.rs:233-269 */
        uint32_t ops = p[0];
        for (uint32_t i = 0; i < ops; i++) {
            const uint64_t *r = w + 5 * (size_t)i;
            uint64_t x = r[0], y = r[1], bi = r[2], res = r[3], bo = r[4];
            uint64_t initial = glo_sub(glo_sub(x, y), bi);
            out[k++] = glo_sub(res, glo_add(initial, glo_mul(bo, 1ULL << 32)));
            uint64_t comb = 0;
            for (int j = 15; j >= 0; j--) {
                uint64_t limb = w[5 * (size_t)ops + 16 * (size_t)i + (size_t)j];
                out[k++] = range_product(limb, 4);
                comb = glo_add(glo_mul(comb, 4), limb);
            }
            out[k++] = glo_sub(comb, res);
            out[k++] = glo_mul(bo, glo_sub(1, bo));
        }
        return k;
    }
    case GLO_GATE_U32_RANGE_CHECK: { /* u32/src/gates/range_check_u32.rs:89-111 */
        uint32_t nl = p[0];
        for (uint32_t i = 0; i < nl; i++) {
            const uint64_t *aux = w + nl + 16 * (size_t)i;
            out[k++] = glo_sub(reduce_with_powers_base(aux, 16, 4), w[i]);
            for (int j = 0; j < 16; j++) out[k++] = range_product(aux[j], 4);
        }
        return k;
    }
    case GLO_GATE_COMPARISON: { /* u32/src/gates/comparison.rs:325-402 */
        uint32_t nb = p[0], nc = p[1], cb = (nb + nc - 1) / nc;
        const uint64_t *first = w + 4, *second = w + 4 + nc;
        out[k++] = glo_sub(reduce_with_powers_base(first, nc, 1ULL << cb), w[0]);
        out[k++] = glo_sub(reduce_with_powers_base(second, nc, 1ULL << cb), w[1]);
        uint64_t msd = 0;
        for (uint32_t i = 0; i < nc; i++) {
            out[k++] = range_product(first[i], 1ULL << cb);
            out[k++] = range_product(second[i], 1ULL << cb);
            uint64_t diff = glo_sub(second[i], first[i]);
            uint64_t dummy = w[4 + 2 * nc + i], eq = w[4 + 3 * nc + i], inter = w[4 + 4 * nc + i];
            out[k++] = glo_sub(glo_mul(diff, dummy), glo_sub(1, eq));
            out[k++] = glo_mul(eq, diff);
            out[k++] = glo_sub(inter, glo_mul(eq, msd));
            msd = glo_add(inter, glo_mul(glo_sub(1, eq), diff));
        }
        out[k++] = glo_sub(w[3], msd);
        const uint64_t *bits = w + 4 + 5 * nc;
        for (uint32_t i = 0; i <= cb; i++) out[k++] = glo_mul(bits[i], glo_sub(1, bits[i]));
        out[k++] = glo_sub(glo_add(w[3], 1ULL << cb), reduce_with_powers_base(bits, cb + 1, 2));
        out[k++] = glo_sub(w[2], bits[cb]);
        return k;
    }
    case GLO_GATE_RANDOM_ACCESS: { /* gates/random_access.rs:409-450 */
        uint32_t bits_n = p[0], copies = p[1], extra = p[2], vs = 1u << bits_n;
        size_t routed = (size_t)(2 + vs) * copies + extra;
        uint64_t items[256];
        if (vs > 256) return -1;
        for (uint32_t c = 0; c < copies; c++) {
            size_t o = (size_t)(2 + vs) * c;
            uint64_t idx = w[o], claimed = w[o + 1];
            const uint64_t *bits = w + routed + (size_t)c * bits_n;
            for (uint32_t i = 0; i < vs; i++) items[i] = w[o + 2 + i];
            for (uint32_t i = 0; i < bits_n; i++) out[k++] = glo_mul(bits[i], glo_sub(bits[i], 1));
            uint64_t rec = 0;
            for (uint32_t i = bits_n; i-- > 0;) rec = glo_add(glo_add(rec, rec), bits[i]);
            out[k++] = glo_sub(rec, idx);
            uint32_t len = vs;
            for (uint32_t b = 0; b < bits_n; b++) {
                for (uint32_t q = 0; q < len / 2; q++) items[q] = glo_add(items[2 * q], glo_mul(bits[b], glo_sub(items[2 * q + 1], items[2 * q])));
                len /= 2;
            }
            out[k++] = glo_sub(items[0], claimed);
        }
        for (uint32_t i = 0; i < extra; i++) out[k++] = glo_sub(consts[i], w[(size_t)(2 + vs) * copies + i]);
        return k;
    }
    case GLO_GATE_POSEIDON:
        return poseidon_gate(w, out);
    case GLO_GATE_ARITHMETIC_EXTENSION: /* gates/arithmetic_extension.rs:129-147: wires 4*D*i + {0, D, 2D, 3D} */
        for (uint32_t i = 0; i < p[0]; i++) {
            e2 m0 = wext(w, 8 * (size_t)i), m1 = wext(w, 8 * (size_t)i + 2), ad = wext(w, 8 * (size_t)i + 4), o = wext(w, 8 * (size_t)i + 6);
            e2 computed = e2_add(e2_scalar(e2_mul(m0, m1), consts[0]), e2_scalar(ad, consts[1]));
            e2 d = e2_sub(o, computed);
            out[k++] = d.a;
            out[k++] = d.b;
        }
        return k;
    case GLO_GATE_MUL_EXTENSION: /* gates/multiplication_extension.rs:122-137: wires 3*D*i + {0, D, 2D} */
        for (uint32_t i = 0; i < p[0]; i++) {
            e2 m0 = wext(w, 6 * (size_t)i), m1 = wext(w, 6 * (size_t)i + 2), o = wext(w, 6 * (size_t)i + 4);
            e2 d = e2_sub(o, e2_scalar(e2_mul(m0, m1), consts[0]));
            out[k++] = d.a;
            out[k++] = d.b;
        }
        return k;
    case GLO_GATE_REDUCING: { /* gates/reducing.rs:160-181: output 0..D, alpha D..2D, old_acc 2D..3D, coeffs 3D.., accs after them (the
                                 last acc IS the output) */
        uint32_t nc = p[0];
        e2 alpha = wext(w, 2), acc = wext(w, 4);
        size_t start_coeffs = 6, start_accs = 6 + nc;
        for (uint32_t i = 0; i < nc; i++) {
            e2 next = i == nc - 1 ? wext(w, 0) : wext(w, start_accs + 2 * (size_t)i);
            e2 t = e2_sub(e2_add(e2_mul(acc, alpha), e2_make(w[start_coeffs + i], 0)), next);
            out[k++] = t.a;
            out[k++] = t.b;
            acc = next;
        }
        return k;
    }
    case GLO_GATE_REDUCING_EXTENSION: { /* gates/reducing_extension.rs:157-178: coeffs are extension elements */
        uint32_t nc = p[0];
        e2 alpha = wext(w, 2), acc = wext(w, 4);
        size_t start_coeffs = 6, start_accs = 6 + 2 * (size_t)nc;
        for (uint32_t i = 0; i < nc; i++) {
            e2 next = i == nc - 1 ? wext(w, 0) : wext(w, start_accs + 2 * (size_t)i);
            e2 t = e2_sub(e2_add(e2_mul(acc, alpha), wext(w, start_coeffs + 2 * (size_t)i)), next);
            out[k++] = t.a;
            out[k++] = t.b;
            acc = next;
        }
        return k;
    }
    case GLO_GATE_EXPONENTIATION: { /* gates/exponentiation.rs:185-219: base 0, power bits 1..1+n (little endian), output 1+n,
                                       intermediate values 2+n.. */
        uint32_t n = p[0];
        uint64_t base = w[0], output = w[1 + n];
        const uint64_t *bits = w + 1, *inter = w + 2 + n;
        for (uint32_t i = 0; i < n; i++) {
            uint64_t prev = i == 0 ? 1 : glo_mul(inter[i - 1], inter[i - 1]);
            uint64_t cur_bit = bits[n - 1 - i]; /* power bits are in little-endian order, but we process them in big-endian order */
            uint64_t not_cur_bit = glo_sub(1, cur_bit);
            uint64_t computed = glo_mul(prev, glo_add(glo_mul(cur_bit, base), not_cur_bit));
            out[k++] = glo_sub(computed, inter[i]);
        }
        out[k++] = glo_sub(output, inter[n - 1]);
        return k;
    }
    case GLO_GATE_POSEIDON_MDS: { /* gates/poseidon_mds.rs:184-204: inputs i*D.., outputs (12+i)*D..; mds_layer_algebra :67-110 over F_p^2 */
        for (int r = 0; r < 12; r++) {
            e2 acc = e2_scalar(wext(w, 2 * (size_t)r), POSEIDON_MDS_DIAG[r]);
            for (int i = 0; i < 12; i++) acc = e2_add(acc, e2_scalar(wext(w, 2 * (size_t)((i + r) % 12)), POSEIDON_MDS_CIRC[i]));
            e2 d = e2_sub(wext(w, 2 * (size_t)(12 + r)), acc);
            out[k++] = d.a;
            out[k++] = d.b;
        }
        return k;
    }
    case GLO_GATE_LOW_DEGREE_INTERPOLATION:
        return interpolation_gate(1, p[0], w, out);
    case GLO_GATE_HIGH_DEGREE_INTERPOLATION:
        return interpolation_gate(0, p[0], w, out);
    default:
        return -1;
    }
}

/* Interpolation gates. Wire layout of both (gates/interpolation.rs:19-76): coset shift at 0; the values at the 2^bits points from 1
 * (D wires each); evaluation point, evaluation value, then the coefficients (D each). */
static e2 poly_eval_e2(const e2 *coeffs, size_t n, e2 x) {
    e2 acc = e2_make(0, 0);
    for (size_t i = n; i-- > 0;) acc = e2_add(e2_mul(acc, x), coeffs[i]);
    return acc;
}
static int interpolation_gate(int low_degree, unsigned subgroup_bits, const uint64_t *w, uint64_t *out) {
    size_t np = (size_t)1 << subgroup_bits;
    if (np > 64) return -1;
    int k = 0;
    uint64_t shift = w[0];
    size_t start_values = 1, wires_eval_point = 1 + 2 * np, wires_eval_value = wires_eval_point + 2, start_coeffs = wires_eval_value + 2;
    e2 coeffs[64];
    for (size_t i = 0; i < np; i++) coeffs[i] = wext(w, start_coeffs + 2 * i);
    uint64_t g = glo_primitive_root_of_unity(subgroup_bits);
    if (!low_degree) {
        /* gates/high_degree_interpolation.rs:119-147 (eval_unfiltered_base_one): interpolant = coeffs; coset = shift * g^i;
         * for each point: value - interpolant.eval_base(point); then evaluation_value - interpolant.eval(evaluation_point) */
        uint64_t pt = shift;
        for (size_t i = 0; i < np; i++) {
            e2 computed = poly_eval_e2(coeffs, np, e2_make(pt, 0));
            e2 d = e2_sub(wext(w, start_values + 2 * i), computed);
            out[k++] = d.a;
            out[k++] = d.b;
            pt = glo_mul(pt, g);
        }
        e2 d = e2_sub(wext(w, wires_eval_value), poly_eval_e2(coeffs, np, wext(w, wires_eval_point)));
        out[k++] = d.a;
        out[k++] = d.b;
        return k;
    }
    /* gates/low_degree_interpolation.rs:356-404 (eval_unfiltered_base_one). Extra wires behind the coefficients (:51-67):
     * powers_shift(i) = shift^i for i in 2..np-1 (np - 2 base wires; i = 1 is the shift wire itself), then
     * powers_evaluation_point(i) for i in 2..np-1 (np - 2 extension elements; i = 1 is the evaluation point itself). */
    size_t end_coeffs = start_coeffs + 2 * np;
    uint64_t powers_shift[64]; /* index i = shift^i after the insert(0, ONE) of :371 */
    powers_shift[0] = 1;
    powers_shift[1] = shift;
    for (size_t i = 2; i < np; i++) powers_shift[i] = w[end_coeffs + i - 2];
    for (size_t i = 1; i + 1 < np; i++) out[k++] = glo_sub(glo_mul(powers_shift[i], shift), powers_shift[i + 1]); /* :368-370 */
    e2 altered[64]; /* altered_coeffs[i] = c_i * shift^i: altered(w^i) = original(shift * w^i) */
    for (size_t i = 0; i < np; i++) altered[i] = e2_scalar(coeffs[i], powers_shift[i]);
    uint64_t pt = 1; /* F::two_adic_subgroup(subgroup_bits) */
    for (size_t i = 0; i < np; i++) {
        e2 computed = poly_eval_e2(altered, np, e2_make(pt, 0));
        e2 d = e2_sub(wext(w, start_values + 2 * i), computed);
        out[k++] = d.a;
        out[k++] = d.b;
        pt = glo_mul(pt, g);
    }
    e2 epp[64]; /* index i = evaluation_point^i */
    epp[0] = e2_make(1, 0);
    epp[1] = wext(w, wires_eval_point);
    for (size_t i = 2; i < np; i++) epp[i] = wext(w, end_coeffs + np - 2 + 2 * (i - 2));
    for (size_t i = 1; i + 1 < np; i++) { /* :392-397 */
        e2 d = e2_sub(e2_mul(epp[i], epp[1]), epp[i + 1]);
        out[k++] = d.a;
        out[k++] = d.b;
    }
    e2 computed = coeffs[0]; /* interpolant.eval_with_powers (field/src/polynomial/mod.rs): c_0 + sum c_i * point^i, the ORIGINAL coefficients */
    for (size_t i = 1; i < np; i++) computed = e2_add(computed, e2_mul(epp[i], coeffs[i]));
    e2 d = e2_sub(wext(w, wires_eval_value), computed);
    out[k++] = d.a;
    out[k++] = d.b;
    return k;
}

/* gates/gate.rs:261-268 compute_filter */
static uint64_t compute_filter(uint32_t row, uint32_t a, uint32_t b, uint64_t s, int many_selectors) {
    uint64_t f = 1;
    for (uint32_t i = a; i < b; i++)
        if (i != row) f = glo_mul(f, glo_sub(i, s));
    if (many_selectors) f = glo_mul(f, glo_sub(UNUSED_SELECTOR, s));
    return f;
}

/* plonk/vanishing_poly.rs:267-306 over Gate::eval_filtered_base_batch (gates/gate.rs:109-150): the gate sees the constants
 * AFTER the selector prefix (vars.remove_prefix(num_selectors)) */
void glo_evaluate_gate_constraints(const glo_circuit_desc *c, const uint64_t *local_constants, const uint64_t *local_wires,
                                   const uint64_t pih[4], uint64_t *out) {
    uint64_t tmp[1024];
    for (uint32_t k = 0; k < c->num_gate_constraints; k++) out[k] = 0;
    for (uint32_t row = 0; row < c->num_gates; row++) {
        const glo_gate *g = &c->gates[row];
        uint32_t si = g->selector_index;
        uint64_t filt = compute_filter(row, c->group_bounds[2 * si], c->group_bounds[2 * si + 1], local_constants[si], c->num_selectors > 1);
        int nk = glo_gate_constraints(g, local_constants + c->num_selectors, local_wires, pih, tmp);
        for (int k = 0; k < nk; k++) out[k] = glo_add(out[k], glo_mul(filt, tmp[k]));
    }
}

/* ------------------------------------------------------------------ commitment (PolynomialBatch) */
typedef struct {
    size_t n_polys, n, n_ext, salt, leaf_len;
    unsigned cap_height;
    uint64_t *coeffs;  /* [n_polys][n]   PolynomialBatch.polynomials */
    uint64_t *leaves;  /* [n_ext][leaf_len] merkle_tree.leaves */
    uint64_t *digests; /* 4 * 2 (n_ext - 2^h) */
    uint64_t *cap;     /* 4 * 2^h */
} commit_t;

static void commit_free(commit_t *c) {
    free(c->coeffs);
    free(c->leaves);
    free(c->digests);
    free(c->cap);
    memset(c, 0, sizeof *c);
}

/* PolynomialBatch::from_coeffs (fri/oracle.rs:911-977): lde_values :979-1004 (salt columns appended iff blinding) ->
 * transpose :942 -> reverse_index_bits_in_place :952 -> MerkleTree::new :962-966. Takes ownership of `coeffs`.
 * The LDE is produced a block of columns at a time and scattered straight into the leaf rows (leaf j = point
 * reverse_bits(j)), so that no second full-size matrix exists. salt: [SALT_SIZE][n_ext] in leaf order, or NULL. */
static int commit_from_coeffs(commit_t *c, uint64_t *coeffs, size_t n_polys, size_t n, unsigned rate_bits, unsigned cap_height,
                              const uint64_t *salt, int threads) {
    memset(c, 0, sizeof *c);
    c->n_polys = n_polys, c->n = n, c->n_ext = n << rate_bits, c->salt = salt ? SALT_SIZE : 0, c->leaf_len = n_polys + c->salt;
    c->cap_height = cap_height;
    c->coeffs = coeffs;
    size_t n_ext = c->n_ext, L = c->leaf_len;
    unsigned lg = log2_strict_sz(n_ext);
    if (cap_height > lg) return -1;
    size_t blk = (size_t)(threads > 0 ? threads : 1) * 2;
    if (blk > n_polys) blk = n_polys ? n_polys : 1;
    c->leaves = (uint64_t *)malloc(n_ext * L * 8);
    uint64_t *tmp = (uint64_t *)malloc(blk * n_ext * 8);
    c->digests = (uint64_t *)malloc((2 * (n_ext - ((size_t)1 << cap_height)) * 4 + 4) * 8);
    c->cap = (uint64_t *)malloc(((size_t)4 << cap_height) * 8);
    if (!c->leaves || !tmp || !c->digests || !c->cap) {
        free(tmp);
        return -2;
    }
    void *rt = glo_root_table_new(n_ext);
    for (size_t p0 = 0; p0 < n_polys; p0 += blk) {
        size_t nb = n_polys - p0 < blk ? n_polys - p0 : blk;
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads > 0 ? threads : 1)
        for (size_t q = 0; q < nb; q++) glo_coset_lde_with_table(coeffs + (p0 + q) * n, n, rate_bits, COSET_SHIFT, tmp + q * n_ext, rt);
#pragma omp parallel for schedule(static) num_threads(threads > 0 ? threads : 1)
        for (size_t i = 0; i < n_ext; i++) {
            size_t src = glo_reverse_bits(i, lg);
            for (size_t q = 0; q < nb; q++) c->leaves[i * L + p0 + q] = tmp[q * n_ext + src];
        }
    }
    glo_root_table_free(rt);
    free(tmp);
    if (salt)
#pragma omp parallel for schedule(static) num_threads(threads > 0 ? threads : 1)
        for (size_t i = 0; i < n_ext; i++)
            for (size_t k = 0; k < SALT_SIZE; k++) c->leaves[i * L + n_polys + k] = salt[k * n_ext + i];
    return glo_merkle_tree(c->leaves, n_ext, L, cap_height, c->digests, c->cap, threads);
}

/* PolynomialBatch::from_values (fri/oracle.rs:709-731): ifft per column, then from_coeffs. `values` is copied. */
static int commit_from_values(commit_t *c, const uint64_t *values, size_t n_polys, size_t n, unsigned rate_bits, unsigned cap_height,
                              const uint64_t *salt, int threads) {
    uint64_t *cf = (uint64_t *)malloc(n_polys * n * 8 + 8);
    if (!cf) return -2;
    memcpy(cf, values, n_polys * n * 8);
    void *rt = glo_root_table_new(n);
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads > 0 ? threads : 1)
    for (size_t p = 0; p < n_polys; p++) glo_ifft_with_table(cf + p * n, n, rt);
    glo_root_table_free(rt);
    return commit_from_coeffs(c, cf, n_polys, n, rate_bits, cap_height, salt, threads);
}

/* ------------------------------------------------------------------ Challenger (iop/challenger.rs:31-160) */
typedef struct {
    uint64_t sponge_state[SPONGE_WIDTH];
    uint64_t input_buffer[SPONGE_RATE];
    uint64_t output_buffer[SPONGE_RATE];
    unsigned n_in, n_out;
} challenger_t;

static void ch_init(challenger_t *c) { memset(c, 0, sizeof *c); }
static void ch_duplexing(challenger_t *c) { /* :131-149 */
    for (unsigned i = 0; i < c->n_in; i++) c->sponge_state[i] = c->input_buffer[i];
    c->n_in = 0;
    glo_poseidon(c->sponge_state);
    for (unsigned i = 0; i < SPONGE_RATE; i++) c->output_buffer[i] = glo_canon(c->sponge_state[i]);
    c->n_out = SPONGE_RATE;
}
static void ch_observe(challenger_t *c, uint64_t e) { /* :43-53 */
    c->n_out = 0;
    c->input_buffer[c->n_in++] = glo_canon(e);
    if (c->n_in == SPONGE_RATE) ch_duplexing(c);
}
static void ch_observe_many(challenger_t *c, const uint64_t *e, size_t n) {
    for (size_t i = 0; i < n; i++) ch_observe(c, e[i]);
}
static void ch_observe_e2s(challenger_t *c, const e2 *e, size_t n) { /* :55-76 */
    for (size_t i = 0; i < n; i++) {
        ch_observe(c, e[i].a);
        ch_observe(c, e[i].b);
    }
}
static uint64_t ch_get(challenger_t *c) { /* :88-98 */
    if (c->n_in != 0 || c->n_out == 0) ch_duplexing(c);
    return c->output_buffer[--c->n_out];
}
static e2 ch_get_e2(challenger_t *c) { /* :114-121 */
    uint64_t a = ch_get(c), b = ch_get(c);
    return e2_make(a, b);
}

/* ------------------------------------------------------------------ the circuit */
typedef struct {
    glo_circuit_desc d;
    uint32_t *arity_bits;
    uint64_t *k_is;
    uint64_t *sigmas; /* [num_routed][n] */
    glo_gate *gates;
    uint32_t *group_bounds;
    commit_t cs; /* constants_sigmas_commitment */
    uint64_t digest[4];
} circuit_t;

/* Hasher::hash_pad (plonk/config.rs:44-52) of the empty domain separator */
static void hash_pad_empty(uint64_t out[4]) {
    uint64_t padded[SPONGE_WIDTH];
    size_t len = 0;
    padded[len++] = 1;
    while ((len + 1) % SPONGE_WIDTH != 0) padded[len++] = 0;
    padded[len++] = 1;
    glo_hash_no_pad(padded, len, out);
}

void *glo_circuit_new(const glo_circuit_desc *desc, int threads) {
    circuit_t *c = (circuit_t *)calloc(1, sizeof *c);
    if (!c) return NULL;
    c->d = *desc;
    size_t n = (size_t)1 << desc->degree_bits;
    c->arity_bits = (uint32_t *)malloc((desc->num_reductions + 1) * sizeof(uint32_t));
    memcpy(c->arity_bits, desc->reduction_arity_bits, desc->num_reductions * sizeof(uint32_t));
    c->k_is = (uint64_t *)malloc(desc->num_routed_wires * 8 + 8);
    memcpy(c->k_is, desc->k_is, desc->num_routed_wires * 8);
    c->sigmas = (uint64_t *)malloc((size_t)desc->num_routed_wires * n * 8 + 8);
    memcpy(c->sigmas, desc->sigmas, (size_t)desc->num_routed_wires * n * 8);
    c->gates = (glo_gate *)malloc((desc->num_gates + 1) * sizeof(glo_gate));
    memcpy(c->gates, desc->gates, desc->num_gates * sizeof(glo_gate));
    c->group_bounds = (uint32_t *)malloc((2 * desc->num_selectors + 1) * sizeof(uint32_t));
    memcpy(c->group_bounds, desc->group_bounds, 2 * desc->num_selectors * sizeof(uint32_t));
    c->d.reduction_arity_bits = c->arity_bits, c->d.k_is = c->k_is, c->d.sigmas = c->sigmas, c->d.gates = c->gates;
    c->d.group_bounds = c->group_bounds, c->d.constants = NULL, c->d.circuit_digest = NULL;
    /* constants_sigmas_commitment (circuit_builder.rs:868-880): constants first, then sigmas; never blinded */
    size_t np = (size_t)desc->num_constants + desc->num_routed_wires;
    uint64_t *vals = (uint64_t *)malloc(np * n * 8 + 8);
    memcpy(vals, desc->constants, (size_t)desc->num_constants * n * 8);
    memcpy(vals + (size_t)desc->num_constants * n, desc->sigmas, (size_t)desc->num_routed_wires * n * 8);
    int rc = commit_from_values(&c->cs, vals, np, n, desc->rate_bits, desc->cap_height, NULL, threads);
    free(vals);
    if (rc != 0) {
        glo_circuit_free(c);
        return NULL;
    }
    if (desc->circuit_digest)
        memcpy(c->digest, desc->circuit_digest, 32);
    else { /* circuit_builder.rs:915-927 */
        size_t cap_len = (size_t)4 << desc->cap_height;
        uint64_t *parts = (uint64_t *)malloc((cap_len + 5) * 8);
        memcpy(parts, c->cs.cap, cap_len * 8);
        hash_pad_empty(parts + cap_len);
        parts[cap_len + 4] = desc->degree_bits;
        glo_hash_no_pad(parts, cap_len + 5, c->digest);
        free(parts);
    }
    for (int i = 0; i < 4; i++) c->digest[i] = glo_canon(c->digest[i]);
    return c;
}

void glo_circuit_free(void *circuit) {
    circuit_t *c = (circuit_t *)circuit;
    if (!c) return;
    free(c->arity_bits);
    free(c->k_is);
    free(c->sigmas);
    free(c->gates);
    free(c->group_bounds);
    commit_free(&c->cs);
    free(c);
}

void glo_circuit_info(const void *circuit, uint64_t digest[4], uint64_t *cap) {
    const circuit_t *c = (const circuit_t *)circuit;
    if (digest) memcpy(digest, c->digest, 32);
    if (cap)
        for (size_t i = 0; i < ((size_t)4 << c->d.cap_height); i++) cap[i] = glo_canon(c->cs.cap[i]);
}

void glo_bytes_free(uint8_t *p) { free(p); }

/* ------------------------------------------------------------------ permutation argument */
/* util/partial_products.rs:41-48 */
static uint32_t num_partial_products(uint32_t n, uint32_t max_degree) { return (n + max_degree - 1) / max_degree - 1; }

/* all_wires_permutation_partial_products (plonk/prover.rs:702-786) in the order the prover commits them (:106-117):
 * out [nch * (1 + num_prods)][n], every Z first, then the partial products challenge-major. */
static void zs_partial_products(const circuit_t *c, const uint64_t *wires, const uint64_t *betas, const uint64_t *gammas, uint64_t *out,
                                int threads) {
    const glo_circuit_desc *d = &c->d;
    size_t n = (size_t)1 << d->degree_bits;
    uint32_t nr = d->num_routed_wires, deg = d->quotient_degree_factor, nch = d->num_challenges;
    uint32_t num_prods = num_partial_products(nr, deg), nchunks = num_prods + 1;
    uint64_t w = glo_primitive_root_of_unity(d->degree_bits);
    uint64_t *chunks = (uint64_t *)malloc(n * nchunks * 8);
    for (uint32_t ch = 0; ch < nch; ch++) {
        uint64_t beta = betas[ch], gamma = gammas[ch];
        /* quotient_chunk_products per subgroup row (prover.rs:744-770) */
#pragma omp parallel for schedule(static) num_threads(threads > 0 ? threads : 1)
        for (size_t i = 0; i < n; i++) {
            uint64_t x = glo_exp(w, i);
            for (uint32_t k = 0; k < nchunks; k++) {
                uint64_t prod = 1;
                for (uint32_t j = k * deg; j < nr && j < (k + 1) * deg; j++) {
                    uint64_t wv = wires[(size_t)j * n + i];
                    uint64_t num = glo_add(glo_add(wv, glo_mul(beta, glo_mul(d->k_is[j], x))), gamma);
                    uint64_t den = glo_add(glo_add(wv, glo_mul(beta, c->sigmas[(size_t)j * n + i])), gamma);
                    prod = glo_mul(prod, glo_mul(num, glo_inverse(den)));
                }
                chunks[i * nchunks + k] = prod;
            }
        }
        /* partial_products_and_z_gx row by row, the last slot swapped for Z(x) (prover.rs:772-781) */
        uint64_t z_x = 1;
        for (size_t i = 0; i < n; i++) {
            uint64_t acc = z_x;
            for (uint32_t k = 0; k < nchunks; k++) {
                acc = glo_mul(acc, chunks[i * nchunks + k]);
                if (k < num_prods) out[((size_t)nch + (size_t)ch * num_prods + k) * n + i] = glo_canon(acc);
            }
            out[(size_t)ch * n + i] = glo_canon(z_x);
            z_x = acc;
        }
    }
    free(chunks);
}

/* ------------------------------------------------------------------ quotient */
/* compute_quotient_polys (plonk/prover.rs:790-1034) with eval_vanishing_poly_base_batch (plonk/vanishing_poly.rs:100-226)
 * point by point. out [nch][n << qdb] receives the coefficients. */
static void compute_quotient_polys(const circuit_t *c, const uint64_t pih[4], const commit_t *wires, const commit_t *zs, const uint64_t *betas,
                                   const uint64_t *gammas, const uint64_t *alphas, uint64_t *out, int threads) {
    const glo_circuit_desc *d = &c->d;
    uint32_t nch = d->num_challenges, nr = d->num_routed_wires, qdf = d->quotient_degree_factor, ngc = d->num_gate_constraints;
    unsigned qdb = log2_ceil_u(qdf);
    size_t step = (size_t)1 << (d->rate_bits - qdb), next_step = (size_t)1 << qdb;
    size_t n = (size_t)1 << d->degree_bits, lde_size = n << qdb, rate = (size_t)1 << qdb;
    unsigned bits = d->degree_bits + d->rate_bits;
    uint32_t num_prods = num_partial_products(nr, qdf), nchunks = num_prods + 1;
    uint64_t w = glo_primitive_root_of_unity(d->degree_bits + qdb);
    /* ZeroPolyOnCoset::new(degree_bits, qdb) (field/src/zero_poly_coset.rs:20-33) */
    uint64_t zh_evals[256], zh_inv[256];
    uint64_t g_pow_n = glo_exp(COSET_SHIFT, n), v = glo_primitive_root_of_unity(qdb), vp = 1;
    for (size_t k = 0; k < rate; k++) {
        zh_evals[k] = glo_sub(glo_mul(g_pow_n, vp), 1);
        zh_inv[k] = glo_inverse(zh_evals[k]);
        vp = glo_mul(vp, v);
    }
    uint64_t n_field = (uint64_t)n;
    size_t nterms = (size_t)nch + (size_t)nch * nchunks + ngc;
#pragma omp parallel num_threads(threads > 0 ? threads : 1)
    {
        uint64_t *terms = (uint64_t *)malloc((nterms + 1) * 8);
#pragma omp for schedule(dynamic, 32) /* BATCH_SIZE = 32 points per task (prover.rs:788, 884) */
        for (size_t i = 0; i < lde_size; i++) {
            uint64_t x = glo_mul(COSET_SHIFT, glo_exp(w, i)); /* shifted_x, prover.rs:903 */
            size_t i_next = (i + next_step) % lde_size;
            /* get_lde_values(i, step) = leaves[reverse_bits(i * step, lde_bits)] (fri/oracle.rs:1007-1018), salt stripped */
            const uint64_t *cs = c->cs.leaves + glo_reverse_bits(i * step, bits) * c->cs.leaf_len;
            const uint64_t *lw = wires->leaves + glo_reverse_bits(i * step, bits) * wires->leaf_len;
            const uint64_t *zpp = zs->leaves + glo_reverse_bits(i * step, bits) * zs->leaf_len;
            const uint64_t *next_zs = zs->leaves + glo_reverse_bits(i_next * step, bits) * zs->leaf_len;
            const uint64_t *s_sigmas = cs + d->num_constants, *pp = zpp + nch;
            uint64_t zh = zh_evals[i % rate];
            uint64_t l_0_x = glo_mul(zh, glo_inverse(glo_mul(n_field, glo_sub(x, 1)))); /* eval_l_0, zero_poly_coset.rs:57-60 */
            size_t t = 0;
            for (uint32_t ch = 0; ch < nch; ch++) terms[t++] = glo_mul(l_0_x, glo_sub(zpp[ch], 1)); /* vanishing_z_1_terms */
            for (uint32_t ch = 0; ch < nch; ch++) { /* check_partial_products (util/partial_products.rs:52-76) */
                uint64_t beta = betas[ch], gamma = gammas[ch];
                for (uint32_t k = 0; k < nchunks; k++) {
                    uint64_t np_ = 1, dp = 1;
                    for (uint32_t j = k * qdf; j < nr && j < (k + 1) * qdf; j++) {
                        np_ = glo_mul(np_, glo_add(glo_add(lw[j], glo_mul(beta, glo_mul(d->k_is[j], x))), gamma));
                        dp = glo_mul(dp, glo_add(glo_add(lw[j], glo_mul(beta, s_sigmas[j])), gamma));
                    }
                    uint64_t prev = k == 0 ? zpp[ch] : pp[(size_t)ch * num_prods + k - 1];
                    uint64_t next = k == nchunks - 1 ? next_zs[ch] : pp[(size_t)ch * num_prods + k];
                    terms[t++] = glo_sub(glo_mul(prev, np_), glo_mul(next, dp));
                }
            }
            glo_evaluate_gate_constraints(d, cs, lw, pih, terms + t);
            t += ngc;
            uint64_t zinv = zh_inv[i % rate];
            for (uint32_t ch = 0; ch < nch; ch++) { /* reduce_with_powers_multi (plonk_common.rs:97-114) */
                uint64_t acc = 0, alpha = alphas[ch];
                for (size_t q = t; q-- > 0;) acc = glo_mac(terms[q], acc, alpha);
                out[(size_t)ch * lde_size + i] = glo_mul(acc, zinv); /* prover.rs:985-991 */
            }
        }
        free(terms);
    }
    /* transpose + coset_ifft per challenge (prover.rs:1004-1021) */
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads > 0 ? threads : 1)
    for (uint32_t ch = 0; ch < nch; ch++) {
        glo_coset_ifft(out + (size_t)ch * lde_size, lde_size, COSET_SHIFT);
        for (size_t i = 0; i < lde_size; i++) out[(size_t)ch * lde_size + i] = glo_canon(out[(size_t)ch * lde_size + i]);
    }
}

/* ------------------------------------------------------------------ openings */
/* p.to_extension().eval(z): Horner from the top coefficient (field/src/polynomial/mod.rs:161-166) */
static e2 eval_poly_e2(const uint64_t *coeffs, size_t n, e2 z) {
    e2 acc = e2_make(0, 0);
    for (size_t i = n; i-- > 0;) {
        acc = e2_mul(acc, z);
        acc.a = glo_add(acc.a, coeffs[i]);
    }
    return acc;
}
static void eval_commitment(const commit_t *c, e2 z, e2 *out, int threads) { /* plonk/proof.rs:314-320 */
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads > 0 ? threads : 1)
    for (size_t p = 0; p < c->n_polys; p++) {
        e2 r = eval_poly_e2(c->coeffs + p * c->n, c->n, z);
        out[p] = e2_make(glo_canon(r.a), glo_canon(r.b));
    }
}

/* ------------------------------------------------------------------ byte sink (util/serialization.rs:465-560) */
typedef struct {
    uint8_t *p;
    size_t len, cap;
    int oom;
} sink_t;
static void sink_bytes(sink_t *s, const void *src, size_t n) {
    if (s->len + n > s->cap) {
        size_t nc = s->cap ? s->cap * 2 : 4096;
        while (nc < s->len + n) nc *= 2;
        uint8_t *q = (uint8_t *)realloc(s->p, nc);
        if (!q) {
            s->oom = 1;
            return;
        }
        s->p = q, s->cap = nc;
    }
    memcpy(s->p + s->len, src, n);
    s->len += n;
}
static void sink_u8(sink_t *s, uint8_t x) { sink_bytes(s, &x, 1); }
static void sink_field(sink_t *s, uint64_t x) { /* write_field :492-497: canonical, little endian */
    uint64_t c = glo_canon(x);
    uint8_t b[8];
    for (int i = 0; i < 8; i++) b[i] = (uint8_t)(c >> (8 * i));
    sink_bytes(s, b, 8);
}
static void sink_fields(sink_t *s, const uint64_t *x, size_t n) {
    for (size_t i = 0; i < n; i++) sink_field(s, x[i]);
}
static void sink_e2s(sink_t *s, const e2 *x, size_t n) { /* write_field_ext_vec :511-532 */
    for (size_t i = 0; i < n; i++) {
        sink_field(s, x[i].a);
        sink_field(s, x[i].b);
    }
}
static void sink_merkle_proof(sink_t *s, const uint64_t *siblings, unsigned len) { /* write_merkle_proof :573-588 */
    sink_u8(s, (uint8_t)len);
    sink_fields(s, siblings, 4 * (size_t)len);
}

/* ------------------------------------------------------------------ FRI */
typedef struct {
    size_t n_leaves, leaf_len;
    uint64_t *leaves, *digests, *cap;
} fri_tree_t;

/* PolynomialBatch::prove_openings (fri/oracle.rs:1047-1112) + fri_proof (fri/prover.rs:24-70), serialised as
 * write_fri_proof (util/serialization.rs:641-655) into `fri_out`. oracles = [constants_sigmas, wires, zs_partial_products,
 * quotient] (FRI_ORACLES, plonk/plonk_common.rs:20-44). */
static int prove_openings(const circuit_t *c, const commit_t *const oracles[4], e2 zeta, challenger_t *ch, sink_t *fri_out, int threads) {
    const glo_circuit_desc *d = &c->d;
    size_t n = (size_t)1 << d->degree_bits;
    uint32_t nch = d->num_challenges;
    e2 alpha = ch_get_e2(ch);
    /* get_fri_instance (plonk/circuit_data.rs:351-371): every polynomial of the four oracles at zeta, the Zs at g*zeta */
    e2 g_zeta = e2_scalar(zeta, glo_primitive_root_of_unity(d->degree_bits));
    e2 *final_poly = (e2 *)calloc(n + 1, sizeof(e2)); /* index k = coefficient of X^k AFTER the multiplication by X */
    e2 *comp = (e2 *)malloc(n * sizeof(e2));
    if (!final_poly || !comp) return -2;
    size_t final_len = 0; /* length of final_poly before the shift by X */
    for (int batch = 0; batch < 2; batch++) {
        e2 point = batch == 0 ? zeta : g_zeta;
        /* alpha.reduce_polys_base (util/reducing.rs:83-95): sum_j alpha^j poly_j, powers restart at 1 in every batch */
        size_t count = 0;
        const uint64_t *polys[4096];
        for (int oi = 0; oi < (batch == 0 ? 4 : 1); oi++) {
            const commit_t *o = batch == 0 ? oracles[oi] : oracles[2];
            size_t np = batch == 0 ? o->n_polys : nch;
            for (size_t p = 0; p < np; p++) {
                if (count >= 4096) return -1;
                polys[count++] = o->coeffs + p * o->n;
            }
        }
        e2 *powers = (e2 *)malloc(count * sizeof(e2));
        powers[0] = e2_make(1, 0);
        for (size_t j = 1; j < count; j++) powers[j] = e2_mul(powers[j - 1], alpha);
#pragma omp parallel for schedule(static) num_threads(threads > 0 ? threads : 1)
        for (size_t k = 0; k < n; k++) {
            e2 acc = e2_make(0, 0);
            for (size_t j = 0; j < count; j++) acc = e2_add(acc, e2_scalar(powers[j], polys[j][k]));
            comp[k] = acc;
        }
        free(powers);
        /* composition_poly.divide_by_linear(point) (field/src/polynomial/division.rs:75-88): Horner scan from the top, drop the
         * last accumulator (the remainder), reverse */
        /* alpha.shift_poly(&mut final_poly): *= alpha^count (reducing.rs:103-106); final_poly += quotient */
        e2 scale = e2_pow(alpha, count);
        for (size_t k = 0; k < final_len; k++) final_poly[k + 1] = e2_mul(final_poly[k + 1], scale);
        e2 acc = e2_make(0, 0);
        for (size_t k = n; k-- > 0;) {
            acc = e2_add(e2_mul(acc, point), comp[k]);
            if (k > 0) final_poly[k] = e2_add(final_poly[k], acc); /* quotient coefficient k-1, stored at k (times X) */
        }
        if (n - 1 > final_len) final_len = n - 1;
    }
    free(comp);
    /* final_poly.coeffs.insert(0, ZERO) (oracle.rs:1085-1087): done by the indexing above; lde(rate_bits).coset_fft(shift) */
    size_t n_lde = n << d->rate_bits;
    uint64_t *coeffs = (uint64_t *)calloc(2 * n_lde, 8); /* planar: [0..n_lde) first components, [n_lde..) second */
    uint64_t *values = (uint64_t *)malloc(2 * n_lde * 8);
    if (!coeffs || !values) return -2;
    for (size_t k = 0; k < n; k++) coeffs[k] = final_poly[k].a, coeffs[n_lde + k] = final_poly[k].b;
    free(final_poly);
    size_t len = n_lde;
    for (int pl = 0; pl < 2; pl++) {
        memcpy(values + pl * len, coeffs + pl * len, len * 8);
        glo_coset_fft(values + pl * len, len, COSET_SHIFT);
    }
    /* fri_committed_trees (fri/prover.rs:77-120) */
    fri_tree_t *trees = (fri_tree_t *)calloc(d->num_reductions + 1, sizeof(fri_tree_t));
    uint64_t shift = COSET_SHIFT;
    for (uint32_t r = 0; r < d->num_reductions; r++) {
        unsigned ab = d->reduction_arity_bits[r];
        size_t arity = (size_t)1 << ab;
        unsigned lg = log2_strict_sz(len);
        fri_tree_t *t = &trees[r];
        t->n_leaves = len >> ab, t->leaf_len = 2 * arity;
        unsigned cap_h = d->cap_height;
        t->leaves = (uint64_t *)malloc(len * 2 * 8);
        /* reverse_index_bits_in_place(values); chunks(arity).map(flatten) */
#pragma omp parallel for schedule(static) num_threads(threads > 0 ? threads : 1)
        for (size_t i = 0; i < len; i++) {
            size_t src = glo_reverse_bits(i, lg);
            t->leaves[2 * i] = glo_canon(values[src]);
            t->leaves[2 * i + 1] = glo_canon(values[len + src]);
        }
        t->digests = (uint64_t *)malloc((2 * (t->n_leaves - ((size_t)1 << cap_h)) * 4 + 4) * 8);
        t->cap = (uint64_t *)malloc(((size_t)4 << cap_h) * 8);
        if (glo_merkle_tree(t->leaves, t->n_leaves, t->leaf_len, cap_h, t->digests, t->cap, threads) != 0) return -1;
        ch_observe_many(ch, t->cap, (size_t)4 << cap_h);
        sink_fields(fri_out, t->cap, (size_t)4 << cap_h); /* commit_phase_merkle_caps */
        e2 beta = ch_get_e2(ch);
        /* coeffs = chunks_exact(arity).map(reduce_with_powers(chunk, beta)) (plonk_common.rs:116-128) */
        size_t new_len = len >> ab;
        uint64_t *nc = (uint64_t *)malloc(2 * new_len * 8);
#pragma omp parallel for schedule(static) num_threads(threads > 0 ? threads : 1)
        for (size_t k = 0; k < new_len; k++) {
            e2 sum = e2_make(0, 0);
            for (size_t i = arity; i-- > 0;) sum = e2_add(e2_mul(sum, beta), e2_make(coeffs[k * arity + i], coeffs[len + k * arity + i]));
            nc[k] = sum.a, nc[new_len + k] = sum.b;
        }
        free(coeffs);
        coeffs = nc;
        len = new_len;
        shift = glo_exp(shift, arity);
        for (int pl = 0; pl < 2; pl++) {
            memcpy(values + pl * len, coeffs + pl * len, len * 8);
            glo_coset_fft(values + pl * len, len, shift);
        }
    }
    free(values);
    size_t final_coeffs = len >> d->rate_bits; /* coeffs.truncate(len >> rate_bits) */
    e2 *fp = (e2 *)malloc((final_coeffs + 1) * sizeof(e2));
    for (size_t k = 0; k < final_coeffs; k++) fp[k] = e2_make(glo_canon(coeffs[k]), glo_canon(coeffs[len + k]));
    free(coeffs);
    ch_observe_e2s(ch, fp, final_coeffs);
    /* fri_proof_of_work (fri/prover.rs:122-171), SMALLEST witness */
    unsigned min_lz = d->proof_of_work_bits + 0; /* 64 - F::order().bits() = 0 */
    uint64_t inter[SPONGE_WIDTH];
    memcpy(inter, ch->sponge_state, sizeof inter);
    for (unsigned i = 0; i < ch->n_in; i++) inter[i] = ch->input_buffer[i];
    unsigned wpos = ch->n_in;
    uint64_t witness = UINT64_MAX;
    for (uint64_t base = 0; witness == UINT64_MAX; base += 1 << 16) {
        uint64_t found = UINT64_MAX;
#pragma omp parallel for schedule(static) reduction(min : found) num_threads(threads > 0 ? threads : 1)
        for (uint64_t cand = base; cand < base + (1 << 16); cand++) {
            uint64_t st[SPONGE_WIDTH];
            memcpy(st, inter, sizeof st);
            st[wpos] = cand;
            glo_poseidon(st);
            uint64_t resp = glo_canon(st[SPONGE_RATE - 1]);
            unsigned lz = resp == 0 ? 64 : (unsigned)__builtin_clzll(resp);
            if (lz >= min_lz && cand < found) found = cand;
        }
        witness = found;
    }
    ch_observe(ch, witness);
    (void)ch_get(ch); /* pow_response, asserted by the reference */
    /* fri_prover_query_rounds (fri/prover.rs:173-260) */
    uint64_t sib[64 * 4];
    for (uint32_t q = 0; q < d->num_query_rounds; q++) {
        size_t x_index = (size_t)(ch_get(ch) % (uint64_t)n_lde);
        for (int oi = 0; oi < 4; oi++) { /* initial_trees_proof: (leaf, MerkleTree::prove) per oracle; write_fri_initial_proof */
            const commit_t *o = oracles[oi];
            sink_fields(fri_out, o->leaves + x_index * o->leaf_len, o->leaf_len);
            unsigned nl = glo_merkle_prove(o->digests, o->n_ext, o->cap_height, x_index, sib);
            sink_merkle_proof(fri_out, sib, nl);
        }
        for (uint32_t r = 0; r < d->num_reductions; r++) {
            unsigned ab = d->reduction_arity_bits[r];
            const fri_tree_t *t = &trees[r];
            sink_fields(fri_out, t->leaves + (x_index >> ab) * t->leaf_len, t->leaf_len); /* evals = unflatten(tree.get(x_index >> ab)) */
            unsigned nl = glo_merkle_prove(t->digests, t->n_leaves, d->cap_height, x_index >> ab, sib);
            sink_merkle_proof(fri_out, sib, nl);
            x_index >>= ab;
        }
    }
    sink_e2s(fri_out, fp, final_coeffs);
    sink_field(fri_out, witness);
    free(fp);
    for (uint32_t r = 0; r < d->num_reductions; r++) {
        free(trees[r].leaves);
        free(trees[r].digests);
        free(trees[r].cap);
    }
    free(trees);
    return 0;
}

/* ------------------------------------------------------------------ prove() */
static int fail(char *err, size_t err_len, int code, const char *msg) {
    if (err && err_len) snprintf(err, err_len, "%s", msg);
    return code;
}

int glo_prove(const void *circuit, const uint64_t *wires, const uint64_t *public_inputs, uint32_t num_public_inputs, const uint64_t *salts,
              uint8_t **proof, size_t *proof_len, glo_prove_trace *trace, int threads, char *err, size_t err_len) {
    const circuit_t *c = (const circuit_t *)circuit;
    const glo_circuit_desc *d = &c->d;
    if (!c || !wires || !proof || !proof_len) return fail(err, err_len, -1, "null argument");
    if ((d->hiding != 0) != (salts != NULL)) return fail(err, err_len, -1, "salts are given exactly when the circuit is hiding");
    if (d->quotient_degree_factor >= d->num_routed_wires) return fail(err, err_len, -1, "quotient_degree_factor >= num_routed_wires (prover.rs:102-105)");
    size_t n = (size_t)1 << d->degree_bits, n_ext = n << d->rate_bits;
    uint32_t nch = d->num_challenges, qdf = d->quotient_degree_factor;
    unsigned qdb = log2_ceil_u(qdf);
    if (qdb > d->rate_bits) return fail(err, err_len, -1, "constraints of degree higher than the rate are not supported (prover.rs:806-810)");
    size_t cap_len = (size_t)4 << d->cap_height;
    double t0 = now_s(), t_prev = t0, t;
    double stage[8] = {0};
    int rc;
    /* public_inputs_hash (prover.rs:57-58; PoseidonHash::hash_public_inputs = hash_no_pad, hash/poseidon.rs:662-664) */
    uint64_t pih[4];
    glo_hash_no_pad(public_inputs, num_public_inputs, pih);
    for (int i = 0; i < 4; i++) pih[i] = glo_canon(pih[i]);
    commit_t wires_c, zs_c, quot_c;
    memset(&zs_c, 0, sizeof zs_c);
    memset(&quot_c, 0, sizeof quot_c);
    /* wires commitment (prover.rs:77-90) */
    rc = commit_from_values(&wires_c, wires, d->num_wires, n, d->rate_bits, d->cap_height, salts ? salts : NULL, threads);
    if (rc != 0) {
        commit_free(&wires_c);
        return fail(err, err_len, rc == -2 ? -2 : -1, "wires commitment failed");
    }
    t = now_s(), stage[0] = t - t_prev, t_prev = t;
    challenger_t ch;
    ch_init(&ch);
    ch_observe_many(&ch, c->digest, 4); /* prover.rs:94-96 */
    ch_observe_many(&ch, pih, 4);
    ch_observe_many(&ch, wires_c.cap, cap_len);
    uint64_t betas[16], gammas[16], alphas[16];
    if (nch > 16) return fail(err, err_len, -1, "num_challenges > 16");
    for (uint32_t i = 0; i < nch; i++) betas[i] = ch_get(&ch);
    for (uint32_t i = 0; i < nch; i++) gammas[i] = ch_get(&ch);
    /* partial products and Zs (prover.rs:106-117) */
    uint32_t num_prods = num_partial_products(d->num_routed_wires, qdf);
    size_t nzs = (size_t)nch * (1 + num_prods);
    uint64_t *zs_pp = (uint64_t *)malloc(nzs * n * 8);
    if (!zs_pp) return fail(err, err_len, -2, "out of memory");
    zs_partial_products(c, wires, betas, gammas, zs_pp, threads);
    t = now_s(), stage[1] = t - t_prev, t_prev = t;
    rc = commit_from_values(&zs_c, zs_pp, nzs, n, d->rate_bits, d->cap_height, salts ? salts + (size_t)SALT_SIZE * n_ext : NULL, threads);
    if (trace && trace->zs_partial_products) memcpy(trace->zs_partial_products, zs_pp, nzs * n * 8);
    free(zs_pp);
    if (rc != 0) goto fail_commit;
    t = now_s(), stage[2] = t - t_prev, t_prev = t;
    ch_observe_many(&ch, zs_c.cap, cap_len);
    for (uint32_t i = 0; i < nch; i++) alphas[i] = ch_get(&ch);
    /* quotient (prover.rs:136-166) */
    size_t lde_size = n << qdb;
    uint64_t *quotient = (uint64_t *)malloc((size_t)nch * lde_size * 8);
    if (!quotient) goto fail_commit;
    compute_quotient_polys(c, pih, &wires_c, &zs_c, betas, gammas, alphas, quotient, threads);
    if (trace && trace->quotient_polys) memcpy(trace->quotient_polys, quotient, (size_t)nch * lde_size * 8);
    /* trim_to_len(quotient_degree) must only drop zeros; chunks(degree) */
    uint64_t *chunks = (uint64_t *)malloc((size_t)nch * qdf * n * 8 + 8);
    for (uint32_t i = 0; i < nch; i++) {
        for (size_t k = (size_t)qdf * n; k < lde_size; k++)
            if (quotient[(size_t)i * lde_size + k] != 0) {
                free(quotient);
                free(chunks);
                commit_free(&wires_c);
                commit_free(&zs_c);
                return fail(err, err_len, -3, "Quotient has failed, the vanishing polynomial is not divisible by Z_H");
            }
        memcpy(chunks + (size_t)i * qdf * n, quotient + (size_t)i * lde_size, (size_t)qdf * n * 8);
    }
    free(quotient);
    t = now_s(), stage[3] = t - t_prev, t_prev = t;
    rc = commit_from_coeffs(&quot_c, chunks, (size_t)nch * qdf, n, d->rate_bits, d->cap_height, salts ? salts + 2 * (size_t)SALT_SIZE * n_ext : NULL, threads);
    if (rc != 0) goto fail_commit;
    t = now_s(), stage[4] = t - t_prev, t_prev = t;
    ch_observe_many(&ch, quot_c.cap, cap_len);
    e2 zeta = ch_get_e2(&ch);
    {
        e2 zp = zeta; /* zeta.exp_power_of_2(degree_bits) != ONE (prover.rs:188-191) */
        for (uint32_t i = 0; i < d->degree_bits; i++) zp = e2_mul(zp, zp);
        if (e2_is_one(zp)) {
            commit_free(&wires_c);
            commit_free(&zs_c);
            commit_free(&quot_c);
            return fail(err, err_len, -4, "Opening point is in the subgroup.");
        }
    }
    /* OpeningSet::new (plonk/proof.rs:305-334) */
    e2 g_zeta = e2_scalar(zeta, glo_primitive_root_of_unity(d->degree_bits));
    size_t ncs = c->cs.n_polys, nq = quot_c.n_polys;
    e2 *cs_eval = (e2 *)malloc((ncs + d->num_wires + 2 * nzs + nq + 4) * sizeof(e2));
    e2 *w_eval = cs_eval + ncs, *zs_eval = w_eval + d->num_wires, *zs_next = zs_eval + nzs, *q_eval = zs_next + nzs;
    eval_commitment(&c->cs, zeta, cs_eval, threads);
    eval_commitment(&wires_c, zeta, w_eval, threads);
    eval_commitment(&zs_c, zeta, zs_eval, threads);
    eval_commitment(&zs_c, g_zeta, zs_next, threads);
    eval_commitment(&quot_c, zeta, q_eval, threads);
    /* challenger.observe_openings(&openings.to_fri_openings()) (proof.rs:336-356): zeta batch = constants, sigmas, wires, zs,
     * partial products, quotient; then the zs at g*zeta */
    ch_observe_e2s(&ch, cs_eval, ncs);
    ch_observe_e2s(&ch, w_eval, d->num_wires);
    ch_observe_e2s(&ch, zs_eval, nch);
    ch_observe_e2s(&ch, zs_eval + nch, nzs - nch);
    ch_observe_e2s(&ch, q_eval, nq);
    ch_observe_e2s(&ch, zs_next, nch);
    t = now_s(), stage[5] = t - t_prev, t_prev = t;
    /* the proof on the wire: write_proof_with_public_inputs (util/serialization.rs:657-689) */
    sink_t s;
    memset(&s, 0, sizeof s);
    sink_fields(&s, wires_c.cap, cap_len);
    sink_fields(&s, zs_c.cap, cap_len);
    sink_fields(&s, quot_c.cap, cap_len);
    sink_e2s(&s, cs_eval, d->num_constants);                 /* constants */
    sink_e2s(&s, cs_eval + d->num_constants, ncs - d->num_constants); /* plonk_sigmas */
    sink_e2s(&s, w_eval, d->num_wires);                      /* wires */
    sink_e2s(&s, zs_eval, nch);                              /* plonk_zs */
    sink_e2s(&s, zs_next, nch);                              /* plonk_zs_next */
    sink_e2s(&s, zs_eval + nch, nzs - nch);                  /* partial_products */
    sink_e2s(&s, q_eval, nq);                                /* quotient_polys */
    free(cs_eval);
    const commit_t *oracles[4] = {&c->cs, &wires_c, &zs_c, &quot_c};
    rc = prove_openings(c, oracles, zeta, &ch, &s, threads);
    if (rc != 0) {
        free(s.p);
        goto fail_commit;
    }
    for (uint32_t i = 0; i < num_public_inputs; i++) sink_field(&s, public_inputs[i]);
    t = now_s(), stage[6] = t - t_prev, stage[7] = t - t0;
    if (trace) {
        if (trace->betas) memcpy(trace->betas, betas, nch * 8);
        if (trace->gammas) memcpy(trace->gammas, gammas, nch * 8);
        if (trace->alphas) memcpy(trace->alphas, alphas, nch * 8);
        if (trace->zeta) trace->zeta[0] = zeta.a, trace->zeta[1] = zeta.b;
        if (trace->wires_cap) memcpy(trace->wires_cap, wires_c.cap, cap_len * 8);
        if (trace->zs_cap) memcpy(trace->zs_cap, zs_c.cap, cap_len * 8);
        if (trace->quotient_cap) memcpy(trace->quotient_cap, quot_c.cap, cap_len * 8);
        memcpy(trace->stage_seconds, stage, sizeof stage);
    }
    commit_free(&wires_c);
    commit_free(&zs_c);
    commit_free(&quot_c);
    if (s.oom) {
        free(s.p);
        return fail(err, err_len, -2, "out of memory");
    }
    *proof = s.p;
    *proof_len = s.len;
    return 0;
fail_commit:
    commit_free(&wires_c);
    commit_free(&zs_c);
    commit_free(&quot_c);
    return fail(err, err_len, rc == -2 ? -2 : -1, "a commitment failed (cap height above the tree height, or out of memory)");
}
