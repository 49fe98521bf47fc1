/* gl_oracle.c — CPU oracle (TEST INFRASTRUCTURE ONLY; see gl_oracle.h for scope and parity pins).
 *
 * Scalar u64/u128 arithmetic like the reference's generic (non-AVX) path. Each function cites
 * the reference file:line it restates. Threading (OpenMP) follows the reference's rayon split:
 * one task per column for NTT/LDE (fri/oracle.rs:720, 990-997), one task per cap subtree plus
 * fork-join recursion for the tree (hash/merkle_tree.rs:96-99, 232-243).
 */
#define _GNU_SOURCE /* sched_setaffinity for the cpu_baseline harness */
#include "gl_oracle.h"

#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "poseidon_constants.h"

typedef unsigned __int128 u128;

/* ------------------------------------------------------------------ field */

/* goldilocks_field.rs:345-358 reduce128 */
static inline uint64_t reduce128(u128 x) {
    uint64_t x_lo = (uint64_t)x, x_hi = (uint64_t)(x >> 64);
    uint64_t x_hi_hi = x_hi >> 32, x_hi_lo = x_hi & GL_EPSILON;
    uint64_t t0 = x_lo - x_hi_hi;
    if (__builtin_expect(x_lo < x_hi_hi, 0)) t0 -= GL_EPSILON; /* borrow: rare (the reference marks it branch_hint()) */
    uint64_t t1 = x_hi_lo * GL_EPSILON;
    /* add_no_canonicalize_trashing_input :321-326. The carry is taken about half the time on random data;
     * the reference does this with a branch-free add/sbb idiom (x86 asm), and so must a restatement that is
     * also timed as the CPU baseline: as an `if` it mispredicts constantly (7.0 vs 1.5 ns per multiply). */
    uint64_t r;
    uint64_t carry = __builtin_add_overflow(t0, t1, &r);
    return r + ((0 - carry) & GL_EPSILON);
}

uint64_t glo_canon(uint64_t a) { return a >= GL_P ? a - GL_P : a; }

uint64_t glo_add(uint64_t a, uint64_t b) {
    uint64_t s;
    uint64_t over = __builtin_add_overflow(a, b, &s); /* branch-free: taken ~half the time on random data */
    uint64_t s2 = s + ((0 - over) & GL_EPSILON);
    if (__builtin_expect(s2 < s, 0)) s2 += GL_EPSILON; /* double overflow, goldilocks_field.rs:205-216 (rare) */
    return s2;
}

uint64_t glo_sub(uint64_t a, uint64_t b) {
    uint64_t d;
    uint64_t under = __builtin_sub_overflow(a, b, &d);
    uint64_t d2 = d - ((0 - under) & GL_EPSILON);
    if (__builtin_expect(d2 > d, 0)) d2 -= GL_EPSILON; /* double underflow :242-253 (rare) */
    return d2;
}

uint64_t glo_neg(uint64_t a) {
    uint64_t c = glo_canon(a);
    return c == 0 ? 0 : GL_P - c;
}

uint64_t glo_mul(uint64_t a, uint64_t b) { return reduce128((u128)a * (u128)b); }

uint64_t glo_mac(uint64_t acc, uint64_t x, uint64_t y) { return reduce128((u128)acc + (u128)x * (u128)y); }

uint64_t glo_exp(uint64_t base, uint64_t power) {
    uint64_t cur = base, prod = 1;
    while (power) {
        if (power & 1) prod = glo_mul(prod, cur);
        cur = glo_mul(cur, cur);
        power >>= 1;
    }
    return prod;
}

/* The reference uses a binary-GCD "plus-minus" inversion (inversion.rs:66); the inverse of a
 * field element is unique, so the canonical value equals x^(p-2). */
uint64_t glo_inverse(uint64_t a) { return glo_canon(glo_exp(a, GL_P - 2)); }

/* types.rs:227-266 (exp <= TWO_ADICITY branch and the > branch) */
uint64_t glo_inverse_2exp(unsigned exp) {
    const unsigned t = 32;
    if (exp > t) {
        uint64_t inv_t = GL_P - ((GL_P - 1) >> t);
        uint64_t res = inv_t;
        unsigned e = exp - t;
        while (e > t) {
            res = glo_mul(res, inv_t);
            e -= t;
        }
        return glo_mul(res, GL_P - ((GL_P - 1) >> e));
    }
    return GL_P - ((GL_P - 1) >> exp);
}

/* types.rs:268-272; POWER_OF_TWO_GENERATOR goldilocks_field.rs:89 */
uint64_t glo_primitive_root_of_unity(unsigned n_log) {
    uint64_t r = 1753635133440165772ULL;
    for (unsigned i = n_log; i < 32; i++) r = glo_mul(r, r);
    return r;
}

/* ------------------------------------------------------------------ bit reversal */

size_t glo_reverse_bits(size_t n, unsigned num_bits) {
    size_t r = 0;
    for (unsigned i = 0; i < num_bits; i++) r |= ((n >> i) & 1) << (num_bits - 1 - i);
    return r;
}

static unsigned log2_strict(size_t n) {
    unsigned l = 0;
    while (((size_t)1 << l) < n) l++;
    return l;
}

/* util/src/lib.rs:188-237 (semantics: swap v[i] <-> v[rev(i)]) */
void glo_reverse_index_bits_in_place(uint64_t *v, size_t n) {
    unsigned lg = log2_strict(n);
    for (size_t i = 0; i < n; i++) {
        size_t j = glo_reverse_bits(i, lg);
        if (i < j) {
            uint64_t t = v[i];
            v[i] = v[j];
            v[j] = t;
        }
    }
}

void glo_reverse_index_bits_rows_in_place(uint64_t *rows, size_t n_rows, size_t row_len) {
    unsigned lg = log2_strict(n_rows);
    uint64_t *tmp = (uint64_t *)malloc(row_len * sizeof(uint64_t));
    for (size_t i = 0; i < n_rows; i++) {
        size_t j = glo_reverse_bits(i, lg);
        if (i < j) {
            memcpy(tmp, rows + i * row_len, row_len * 8);
            memcpy(rows + i * row_len, rows + j * row_len, row_len * 8);
            memcpy(rows + j * row_len, tmp, row_len * 8);
        }
    }
    free(tmp);
}

/* plonky2/src/util/mod.rs:23-53 */
void glo_transpose(const uint64_t *src, uint64_t *dst, size_t rows, size_t cols) {
    for (size_t i = 0; i < cols; i++)
        for (size_t j = 0; j < rows; j++) dst[i * rows + j] = src[j * cols + i];
}

/* ------------------------------------------------------------------ FFT */

/* fft.rs:15-34. Row s (lg_m = s+1) holds max(2^s, 2) powers of w_{2^(s+1)}. */
typedef struct {
    unsigned lg_n;
    uint64_t **rows;
} root_table_t;

static root_table_t root_table_new(size_t n) {
    root_table_t t;
    t.lg_n = log2_strict(n);
    t.rows = (uint64_t **)calloc(t.lg_n ? t.lg_n : 1, sizeof(uint64_t *));
    uint64_t bases[64];
    uint64_t base = glo_primitive_root_of_unity(t.lg_n);
    if (t.lg_n) bases[0] = base;
    for (unsigned i = 1; i < t.lg_n; i++) {
        base = glo_mul(base, base);
        bases[i] = base;
    }
    for (unsigned lg_m = 1; lg_m <= t.lg_n; lg_m++) {
        size_t half_m = (size_t)1 << (lg_m - 1);
        size_t len = half_m > 2 ? half_m : 2;
        uint64_t b = bases[t.lg_n - lg_m];
        uint64_t *row = (uint64_t *)malloc(len * sizeof(uint64_t));
        uint64_t cur = 1;
        for (size_t i = 0; i < len; i++) {
            row[i] = cur;
            cur = glo_mul(cur, b);
        }
        t.rows[lg_m - 1] = row;
    }
    return t;
}

static void root_table_free(root_table_t *t) {
    for (unsigned i = 0; i < t->lg_n; i++) free(t->rows[i]);
    free(t->rows);
}

size_t glo_fft_root_table_concat(size_t n, uint64_t *out) {
    root_table_t t = root_table_new(n);
    size_t k = 0;
    for (unsigned s = 0; s < t.lg_n; s++) {
        size_t half_m = (size_t)1 << s;
        size_t len = half_m > 2 ? half_m : 2;
        if (out) memcpy(out + k, t.rows[s], len * 8);
        k += len;
    }
    root_table_free(&t);
    return k;
}

/* fft.rs:188-229 fft_classic + the scalar instantiation of fft_classic_simd :107-180 */
static void fft_classic(uint64_t *values, size_t n, unsigned r, const root_table_t *rt) {
    glo_reverse_index_bits_in_place(values, n);
    unsigned lg_n = log2_strict(n);
    if (r > 0) {
        size_t mask = ~(((size_t)1 << r) - 1);
        for (size_t i = 0; i < n; i++) values[i] = values[i & mask];
    }
    for (unsigned lg_half_m = r; lg_half_m < lg_n; lg_half_m++) {
        size_t half_m = (size_t)1 << lg_half_m, m = half_m << 1;
        const uint64_t *omega_table = rt->rows[lg_half_m];
        for (size_t k = 0; k < n; k += m) {
            for (size_t j = 0; j < half_m; j++) {
                uint64_t t = glo_mul(omega_table[j], values[k + half_m + j]);
                uint64_t u = values[k + j];
                values[k + j] = glo_add(u, t);
                values[k + half_m + j] = glo_sub(u, t);
            }
        }
    }
}

static void fft_with_table(uint64_t *v, size_t n, unsigned r, const root_table_t *rt) {
    if (n <= 1) return;
    fft_classic(v, n, r, rt);
}

void glo_fft(uint64_t *v, size_t n, unsigned r) {
    if (n <= 1) return;
    root_table_t rt = root_table_new(n);
    fft_classic(v, n, r, &rt);
    root_table_free(&rt);
}

/* fft.rs:73-103 */
static void ifft_with_table(uint64_t *buffer, size_t n, const root_table_t *rt) {
    unsigned lg_n = log2_strict(n);
    uint64_t n_inv = glo_inverse_2exp(lg_n);
    fft_with_table(buffer, n, 0, rt);
    buffer[0] = glo_mul(buffer[0], n_inv);
    if (n > 1) buffer[n / 2] = glo_mul(buffer[n / 2], n_inv);
    for (size_t i = 1; i < n / 2; i++) {
        size_t j = n - i;
        uint64_t ci = glo_mul(buffer[j], n_inv);
        uint64_t cj = glo_mul(buffer[i], n_inv);
        buffer[i] = ci;
        buffer[j] = cj;
    }
}

void glo_ifft(uint64_t *v, size_t n) {
    root_table_t rt = root_table_new(n);
    ifft_with_table(v, n, &rt);
    root_table_free(&rt);
}

/* polynomial/mod.rs:205-207 (lde = zero pad) + :286-299 (coset_fft_with_options) */
static void coset_lde_with_table(const uint64_t *coeffs, size_t n, unsigned rate_bits, uint64_t shift,
                                 uint64_t *out, const root_table_t *rt_ext) {
    size_t n_ext = n << rate_bits;
    uint64_t r = 1;
    for (size_t i = 0; i < n; i++) {
        out[i] = glo_mul(r, coeffs[i]);
        r = glo_mul(r, shift);
    }
    for (size_t i = n; i < n_ext; i++) out[i] = 0; /* r * 0 */
    fft_with_table(out, n_ext, rate_bits, rt_ext);
}

void glo_coset_lde(const uint64_t *coeffs, size_t n, unsigned rate_bits, uint64_t shift, uint64_t *out) {
    root_table_t rt = root_table_new(n << rate_bits);
    coset_lde_with_table(coeffs, n, rate_bits, shift, out, &rt);
    root_table_free(&rt);
}

void glo_coset_fft(uint64_t *v, size_t n, uint64_t shift) {
    uint64_t r = 1;
    for (size_t i = 0; i < n; i++) {
        v[i] = glo_mul(r, v[i]);
        r = glo_mul(r, shift);
    }
    glo_fft(v, n, 0);
}

/* polynomial/mod.rs:64-77 */
void glo_coset_ifft(uint64_t *v, size_t n, uint64_t shift) {
    glo_ifft(v, n);
    uint64_t s_inv = glo_inverse(shift), r = 1;
    for (size_t i = 0; i < n; i++) {
        v[i] = glo_mul(v[i], r);
        r = glo_mul(r, s_inv);
    }
}

/* The root table as an opaque object, so that a caller with many columns builds it once — fft_root_table is built once per
 * circuit and handed to every transform (circuit_builder.rs:849-851, fri/oracle.rs:709-731). Used by prove_oracle.c. */
void *glo_root_table_new(size_t n) {
    root_table_t *t = (root_table_t *)malloc(sizeof *t);
    *t = root_table_new(n);
    return t;
}
void glo_root_table_free(void *t) {
    if (!t) return;
    root_table_free((root_table_t *)t);
    free(t);
}
void glo_fft_with_table(uint64_t *v, size_t n, unsigned r, const void *t) { fft_with_table(v, n, r, (const root_table_t *)t); }
void glo_ifft_with_table(uint64_t *v, size_t n, const void *t) {
    if (n <= 1) return;
    ifft_with_table(v, n, (const root_table_t *)t);
}
void glo_coset_lde_with_table(const uint64_t *coeffs, size_t n, unsigned rate_bits, uint64_t shift, uint64_t *out, const void *t_ext) {
    coset_lde_with_table(coeffs, n, rate_bits, shift, out, (const root_table_t *)t_ext);
}

/* lde(rate_bits).coset_fft(shift) of several columns, one task per column (fri/oracle.rs:990-997) */
void glo_coset_lde_batch(const uint64_t *coeffs, size_t n_polys, size_t n, unsigned rate_bits, uint64_t shift, uint64_t *out, int n_threads) {
    root_table_t rt = root_table_new(n << rate_bits);
#pragma omp parallel for schedule(dynamic, 1) num_threads(n_threads > 0 ? n_threads : 1)
    for (size_t p = 0; p < n_polys; p++) coset_lde_with_table(coeffs + p * n, n, rate_bits, shift, out + p * (n << rate_bits), &rt);
    root_table_free(&rt);
}

void glo_fft_batch(uint64_t *v, size_t n_polys, size_t n, int inverse, int n_threads) {
    root_table_t rt = root_table_new(n);
#pragma omp parallel for schedule(dynamic, 1) num_threads(n_threads > 0 ? n_threads : 1)
    for (size_t p = 0; p < n_polys; p++) {
        if (inverse)
            ifft_with_table(v + p * n, n, &rt);
        else
            fft_with_table(v + p * n, n, 0, &rt);
    }
    root_table_free(&rt);
}

/* cpu_baseline of bench.py: forward + inverse transforms of `cols_per_thread` columns of length n on each of
 * `n_threads` threads, measured the way the reference's prover meets this work: the root table exists before the
 * timed region (fft_root_table is built once per circuit, circuit_builder.rs:849-851, and handed to every
 * fft_with_options call), each thread allocates and fills its own columns (first touch on its own NUMA node, as rayon
 * workers that produced the witness columns would have), one column per task (oracle.rs:720). The clock runs inside
 * the parallel region between two barriers, so thread start-up, allocation and the table are outside it.
 * Returns seconds for the 2 * n_threads * cols_per_thread transforms; *checksum defeats dead-code elimination and
 * lets the caller see that ifft(fft(x)) == x held (0 = every column came back unchanged). */
/* NUMA node of every CPU from /sys/devices/system/node/node<k>/cpulist (no libnuma in this image); -1 = unknown. */
#define GLO_MAX_CPUS 4096
#define GLO_MAX_NODES 64
static int cpu_nodes(int *node_of_cpu) {
    int n_nodes = 0;
    for (int c = 0; c < GLO_MAX_CPUS; c++) node_of_cpu[c] = -1;
    for (int k = 0; k < GLO_MAX_NODES; k++) {
        char path[96];
        snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", k);
        FILE *f = fopen(path, "r");
        if (!f) continue;
        int lo, hi, got = 0;
        char sep;
        while (fscanf(f, "%d", &lo) == 1) {
            hi = lo;
            if (fscanf(f, "%c", &sep) == 1 && sep == '-') {
                if (fscanf(f, "%d", &hi) != 1) break;
                if (fscanf(f, "%c", &sep) != 1) sep = 0;
            }
            for (int c = lo; c <= hi && c < GLO_MAX_CPUS; c++) node_of_cpu[c] = k, got = 1;
            if (sep != ',') break;
        }
        fclose(f);
        if (got && k + 1 > n_nodes) n_nodes = k + 1;
    }
    return n_nodes;
}

double glo_fft_bench(size_t n, int n_threads, int cols_per_thread, uint64_t seed, uint64_t *checksum) {
    if (n_threads < 1) n_threads = 1;
    if (cols_per_thread < 1) cols_per_thread = 1;
    /* The CPUs this process may use, and their NUMA nodes. Thread t is pinned to a core of its own while there are cores
     * (spread evenly over them, so over both sockets), allocates and fills its columns there (first touch on its own node)
     * and reads the copy of the root table that the first thread of its node built. What made round 2's harness SLOWER
     * beyond 32 threads turned out to be none of this: the GPU box's container has a CPU quota
     * (cgroup cpu.max = 16 CPUs on the pool's boxes) and 256 threads time-slice 16 CPUs' worth of time — bench.py now
     * reads the quota (oracle.cpu_quota) and runs as many threads as it grants. */
    cpu_set_t allowed;
    int cpus[GLO_MAX_CPUS], n_cpus = 0;
    static int node_of_cpu[GLO_MAX_CPUS];
    int n_nodes = cpu_nodes(node_of_cpu);
    if (sched_getaffinity(0, sizeof allowed, &allowed) == 0) {
        /* physical cores first (the lowest-numbered hardware thread of every core), their SMT siblings after them: n_threads
         * <= cores then means one thread per core */
        for (int pass = 0; pass < 2; pass++)
            for (int c = 0; c < CPU_SETSIZE && c < GLO_MAX_CPUS; c++) {
                if (!CPU_ISSET(c, &allowed)) continue;
                char path[112];
                int first = c;
                snprintf(path, sizeof path, "/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list", c);
                FILE *f = fopen(path, "r");
                if (f) {
                    if (fscanf(f, "%d", &first) != 1) first = c;
                    fclose(f);
                }
                if ((first == c) == (pass == 0)) cpus[n_cpus++] = c;
            }
    }
    int n_primary = 0;
    for (int i = 0; i < n_cpus; i++) {
        /* the list is primaries then siblings: count the primaries (a sibling's id is never the first of its list) */
        char path[112];
        int first = cpus[i];
        snprintf(path, sizeof path, "/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list", cpus[i]);
        FILE *f = fopen(path, "r");
        if (f) {
            if (fscanf(f, "%d", &first) != 1) first = cpus[i];
            fclose(f);
        }
        if (first == cpus[i]) n_primary++;
    }
    if (n_nodes < 1) n_nodes = 1;
    root_table_t tables[GLO_MAX_NODES];
    int table_ready[GLO_MAX_NODES];
    for (int k = 0; k < GLO_MAX_NODES; k++) table_ready[k] = 0;
    double t0 = 0, t1 = 0;
    uint64_t bad = 0;
#pragma omp parallel num_threads(n_threads) reduction(| : bad)
    {
        int tid = omp_get_thread_num(), nt = omp_get_num_threads();
        int node = 0;
        if (n_cpus > 0 && !getenv("GLO_FFT_BENCH_NO_PIN")) {
            /* up to one thread per core: spread over the cores (and so over the sockets); more: fill the siblings in order */
            int cpu = nt <= n_primary ? cpus[(int)(((long)tid * n_primary) / nt)] : cpus[tid % n_cpus];
            cpu_set_t one;
            CPU_ZERO(&one);
            CPU_SET(cpu, &one);
            (void)sched_setaffinity(0, sizeof one, &one);
            if (node_of_cpu[cpu] >= 0) node = node_of_cpu[cpu];
        }
#pragma omp barrier
        /* one root table per NUMA node, built by one of the node's threads (fft_root_table is built once per circuit,
         * circuit_builder.rs:849-851; a rayon worker reads it through its socket's caches) */
#pragma omp critical(glo_fft_bench_tables)
        {
            if (!table_ready[node]) {
                tables[node] = root_table_new(n);
                table_ready[node] = 1;
            }
        }
#pragma omp barrier
        const root_table_t *rt = &tables[node];
        /* columns of different threads start 8320 bytes apart modulo the page size: 8 MiB columns that are all page-aligned
         * fall on the same cache sets */
        size_t skew = ((size_t)tid % 61) * 8320 / sizeof(uint64_t);
        uint64_t *v_alloc = (uint64_t *)malloc(((size_t)cols_per_thread * n + skew) * sizeof(uint64_t));
        uint64_t *v = v_alloc + skew;
        uint64_t *ref = (uint64_t *)malloc(n * sizeof(uint64_t));
        uint64_t x = seed + 0x9E3779B97F4A7C15ull * (uint64_t)(tid + 1);
        for (size_t i = 0; i < (size_t)cols_per_thread * n; i++) { /* SplitMix64, rejected to [0, p) */
            uint64_t z;
            do {
                x += 0x9E3779B97F4A7C15ull;
                z = x;
                z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
                z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
                z ^= z >> 31;
            } while (z >= GL_P);
            v[i] = z;
        }
        memcpy(ref, v, n * sizeof(uint64_t));
#pragma omp barrier
#pragma omp master
        t0 = omp_get_wtime();
        for (int c = 0; c < cols_per_thread; c++) {
            fft_with_table(v + (size_t)c * n, n, 0, rt);
            ifft_with_table(v + (size_t)c * n, n, rt);
        }
#pragma omp barrier
#pragma omp master
        t1 = omp_get_wtime();
        for (size_t i = 0; i < n; i++) bad |= glo_canon(v[i]) ^ ref[i];
        free(v_alloc);
        free(ref);
        /* the pool's threads are reused by later parallel regions of this process: give them the whole mask back */
        if (n_cpus > 0) (void)sched_setaffinity(0, sizeof allowed, &allowed);
    }
    for (int k = 0; k < GLO_MAX_NODES; k++)
        if (table_ready[k]) root_table_free(&tables[k]);
    if (checksum) *checksum = bad;
    return t1 - t0;
}

/* ------------------------------------------------------------------ Poseidon */

#define W 12
#define HALF_N_FULL_ROUNDS 4
#define N_PARTIAL_ROUNDS 22

/* unsafe add_canonical_u64 goldilocks_field.rs:152-156 */
static inline uint64_t add_canonical_u64(uint64_t x, uint64_t rhs) {
    uint64_t s = x + rhs;
    return s + (s < x ? GL_EPSILON : 0);
}

/* poseidon.rs:484-493 */
static void constant_layer(uint64_t *s, int round_ctr) {
    for (int i = 0; i < W; i++) s[i] = add_canonical_u64(s[i], POSEIDON_ALL_ROUND_CONSTANTS[i + W * round_ctr]);
}

/* poseidon.rs:522-528 */
static inline uint64_t sbox_monomial(uint64_t x) {
    uint64_t x2 = glo_mul(x, x), x4 = glo_mul(x2, x2), x3 = glo_mul(x, x2);
    return glo_mul(x3, x4);
}

static void sbox_layer(uint64_t *s) {
    for (int i = 0; i < W; i++) s[i] = sbox_monomial(s[i]);
}

/* poseidon.rs:174-194 mds_row_shf + :238-260 mds_layer (u128 row sum, from_noncanonical_u96) */
static void mds_layer(uint64_t *s) {
    uint64_t out[W];
    for (int r = 0; r < W; r++) {
        u128 res = 0;
        for (int i = 0; i < W; i++) res += (u128)s[(i + r) % W] * (u128)POSEIDON_MDS_CIRC[i];
        res += (u128)s[r] * (u128)POSEIDON_MDS_DIAG[r];
        out[r] = reduce128(res); /* from_noncanonical_u96((lo, hi32)) types.rs:347-351 */
    }
    memcpy(s, out, sizeof(out));
}

static void full_rounds(uint64_t *s, int *round_ctr) { /* poseidon.rs:574-584 */
    for (int i = 0; i < HALF_N_FULL_ROUNDS; i++) {
        constant_layer(s, *round_ctr);
        sbox_layer(s);
        mds_layer(s);
        (*round_ctr)++;
    }
}

/* poseidon.rs:312-320 */
static void partial_first_constant_layer(uint64_t *s) {
    for (int i = 0; i < W; i++) s[i] = glo_add(s[i], POSEIDON_FAST_PARTIAL_FIRST_ROUND_CONSTANT[i]);
}

/* poseidon.rs:339-365 */
static void mds_partial_layer_init(uint64_t *s) {
    uint64_t out[W];
    memset(out, 0, sizeof(out));
    out[0] = s[0];
    for (int r = 1; r < W; r++)
        for (int c = 1; c < W; c++) {
            uint64_t t = POSEIDON_FAST_PARTIAL_ROUND_INITIAL_MATRIX[(r - 1) * 11 + (c - 1)];
            out[c] = glo_add(out[c], glo_mul(s[r], t));
        }
    memcpy(s, out, sizeof(out));
}

/* poseidon.rs:34-47 add_u160_u128 / reduce_u160; :400-427 mds_partial_layer_fast */
static void mds_partial_layer_fast(uint64_t *s, int r) {
    u128 lo = 0;
    uint32_t hi = 0;
    for (int i = 1; i < W; i++) {
        u128 t = (u128)s[i] * (u128)POSEIDON_FAST_PARTIAL_ROUND_W_HATS[r * 11 + (i - 1)];
        u128 nl = lo + t;
        hi += nl < lo;
        lo = nl;
    }
    {
        u128 t = (u128)s[0] * (u128)(POSEIDON_MDS_CIRC[0] + POSEIDON_MDS_DIAG[0]);
        u128 nl = lo + t;
        hi += nl < lo;
        lo = nl;
    }
    /* reduce_u160 */
    uint64_t n_lo_hi = (uint64_t)(lo >> 64), n_lo_lo = (uint64_t)lo;
    uint64_t reduced_hi = reduce128(((u128)hi << 64) + n_lo_hi);
    uint64_t d = reduce128(((u128)reduced_hi << 64) + n_lo_lo);
    uint64_t out[W];
    out[0] = d;
    for (int i = 1; i < W; i++) out[i] = glo_mac(s[i], s[0], POSEIDON_FAST_PARTIAL_ROUND_VS[r * 11 + (i - 1)]);
    memcpy(s, out, sizeof(out));
}

static void partial_rounds(uint64_t *s, int *round_ctr) { /* poseidon.rs:587-599 */
    partial_first_constant_layer(s);
    mds_partial_layer_init(s);
    for (int i = 0; i < N_PARTIAL_ROUNDS; i++) {
        s[0] = sbox_monomial(s[0]);
        s[0] = add_canonical_u64(s[0], POSEIDON_FAST_PARTIAL_ROUND_CONSTANTS[i]);
        mds_partial_layer_fast(s, i);
    }
    *round_ctr += N_PARTIAL_ROUNDS;
}

void glo_poseidon(uint64_t state[12]) { /* poseidon.rs:602-616 */
    int rc = 0;
    full_rounds(state, &rc);
    partial_rounds(state, &rc);
    full_rounds(state, &rc);
}

void glo_poseidon_naive(uint64_t state[12]) { /* poseidon.rs:620-640 */
    int rc = 0;
    full_rounds(state, &rc);
    for (int i = 0; i < N_PARTIAL_ROUNDS; i++) {
        constant_layer(state, rc);
        state[0] = sbox_monomial(state[0]);
        mds_layer(state);
        rc++;
    }
    full_rounds(state, &rc);
}

/* hashing.rs:81-108, num_outputs = 4 (<= SPONGE_RATE so no squeeze permutation) */
void glo_hash_no_pad(const uint64_t *in, size_t len, uint64_t out[4]) {
    uint64_t st[W];
    memset(st, 0, sizeof(st));
    for (size_t off = 0; off < len; off += 8) {
        size_t c = len - off < 8 ? len - off : 8;
        memcpy(st, in + off, c * 8); /* overwrite mode; short last chunk keeps old lanes */
        glo_poseidon(st);
    }
    memcpy(out, st, 32);
}

/* plonk/config.rs:56-67 */
void glo_hash_or_noop(const uint64_t *in, size_t len, uint64_t out[4]) {
    if (len <= 4) {
        for (int i = 0; i < 4; i++) out[i] = (size_t)i < len ? glo_canon(in[i]) : 0;
    } else {
        glo_hash_no_pad(in, len, out);
    }
}

/* hashing.rs:65-72 */
void glo_two_to_one(const uint64_t l[4], const uint64_t r[4], uint64_t out[4]) {
    uint64_t st[W];
    memcpy(st, l, 32);
    memcpy(st + 4, r, 32);
    memset(st + 8, 0, 32);
    glo_poseidon(st);
    memcpy(out, st, 32);
}

/* ------------------------------------------------------------------ Merkle tree */

/* merkle_tree.rs:78-105: layout = left subtree || left digest || right digest || right subtree.
 * digests_len counts hashes (4 u64 each). Returns the subtree's root in `root`. */
static void fill_subtree(uint64_t *digests_buf, size_t digests_len, const uint64_t *leaves, size_t n_leaves,
                         size_t leaf_len, uint64_t root[4]) {
    if (digests_len == 0) {
        glo_hash_or_noop(leaves, leaf_len, root);
        return;
    }
    size_t half = digests_len / 2;
    uint64_t *left_buf = digests_buf;                    /* half-1 hashes */
    uint64_t *left_digest = digests_buf + 4 * (half - 1);
    uint64_t *right_digest = digests_buf + 4 * half;
    uint64_t *right_buf = digests_buf + 4 * (half + 1);  /* half-1 hashes */
    uint64_t ld[4], rd[4];
    if (n_leaves >= 1024) {
#pragma omp task shared(ld)
        fill_subtree(left_buf, half - 1, leaves, n_leaves / 2, leaf_len, ld);
#pragma omp task shared(rd)
        fill_subtree(right_buf, half - 1, leaves + (n_leaves / 2) * leaf_len, n_leaves / 2, leaf_len, rd);
#pragma omp taskwait
    } else {
        fill_subtree(left_buf, half - 1, leaves, n_leaves / 2, leaf_len, ld);
        fill_subtree(right_buf, half - 1, leaves + (n_leaves / 2) * leaf_len, n_leaves / 2, leaf_len, rd);
    }
    memcpy(left_digest, ld, 32);
    memcpy(right_digest, rd, 32);
    glo_two_to_one(ld, rd, root);
}

/* merkle_tree.rs:210-244 fill_digests_buf + :283-319 new */
int glo_merkle_tree(const uint64_t *leaves, size_t n_leaves, size_t leaf_len, unsigned cap_height,
                    uint64_t *digests, uint64_t *cap, int n_threads) {
    unsigned lg = log2_strict(n_leaves);
    if (cap_height > lg) return -1;
    size_t len_cap = (size_t)1 << cap_height;
    size_t num_digests = 2 * (n_leaves - len_cap);
    size_t sub_digests = num_digests >> cap_height, sub_leaves = n_leaves >> cap_height;
    (void)n_threads;
#pragma omp parallel num_threads(n_threads > 0 ? n_threads : 1)
#pragma omp single
    for (size_t c = 0; c < len_cap; c++) {
#pragma omp task firstprivate(c)
        fill_subtree(digests + 4 * c * sub_digests, sub_digests, leaves + c * sub_leaves * leaf_len, sub_leaves,
                     leaf_len, cap + 4 * c);
    }
    return 0;
}

/* merkle_tree.rs:392-440 */
unsigned glo_merkle_prove(const uint64_t *digests, size_t n_leaves, unsigned cap_height, size_t leaf_index,
                          uint64_t *siblings) {
    unsigned num_layers = log2_strict(n_leaves) - cap_height;
    size_t num_digests = 2 * (n_leaves - ((size_t)1 << cap_height));
    size_t tree_len = num_digests >> cap_height;
    size_t tree_index = leaf_index >> num_layers;
    const uint64_t *tree = digests + 4 * tree_len * tree_index;
    size_t pair_index = leaf_index & (((size_t)1 << num_layers) - 1);
    for (unsigned i = 0; i < num_layers; i++) {
        size_t parity = pair_index & 1;
        pair_index >>= 1;
        size_t siblings_index = (pair_index << (i + 1)) + ((size_t)1 << i) - 1;
        size_t sibling_index = 2 * siblings_index + (1 - parity);
        memcpy(siblings + 4 * i, tree + 4 * sibling_index, 32);
    }
    return num_layers;
}

/* merkle_proofs.rs:53-80 */
int glo_merkle_verify(const uint64_t *leaf, size_t leaf_len, size_t leaf_index, const uint64_t *cap,
                      const uint64_t *siblings, unsigned num_layers) {
    uint64_t cur[4], nxt[4];
    size_t index = leaf_index;
    glo_hash_or_noop(leaf, leaf_len, cur);
    for (unsigned i = 0; i < num_layers; i++) {
        size_t bit = index & 1;
        index >>= 1;
        if (bit)
            glo_two_to_one(siblings + 4 * i, cur, nxt);
        else
            glo_two_to_one(cur, siblings + 4 * i, nxt);
        memcpy(cur, nxt, 32);
    }
    for (int k = 0; k < 4; k++)
        if (glo_canon(cur[k]) != glo_canon(cap[4 * index + k])) return 0;
    return 1;
}

/* ------------------------------------------------------------------ PolynomialBatch */

/* fri/oracle.rs:911-977 from_coeffs: lde_values (:979-1004) -> transpose (:942) ->
 * reverse_index_bits_in_place (:952) -> MerkleTree::new (:962-966) */
int glo_commit_from_coeffs(const uint64_t *coeffs, size_t n_polys, size_t n, unsigned rate_bits,
                           unsigned cap_height, uint64_t *leaves, uint64_t *digests, uint64_t *cap,
                           int n_threads) {
    size_t n_ext = n << rate_bits;
    if (cap_height > log2_strict(n_ext)) return -1;
    /* The reference materialises the whole LDE [P][n_ext], transposes it and permutes the rows (three full-size matrices in
     * flight). Same values here with one: the LDE is produced a block of columns at a time (one column per task, as rayon
     * does) and every block is transposed straight into its place in the bit-reversed leaf rows — so that the full-width
     * 2^21..2^23-row commitments fit the host memory of a test box. */
    int nt = n_threads > 0 ? n_threads : 1;
    size_t blk = (size_t)nt * 2;
    if (blk > n_polys) blk = n_polys ? n_polys : 1;
    uint64_t *tmp = (uint64_t *)malloc(blk * n_ext * 8);
    uint64_t *lv = leaves ? leaves : (uint64_t *)malloc(n_polys * n_ext * 8);
    if (!tmp || !lv) {
        free(tmp);
        if (!leaves) free(lv);
        return -2;
    }
    root_table_t rt = root_table_new(n_ext);
    const uint64_t shift = 7; /* F::coset_shift() types.rs:431-433 */
    unsigned lg = log2_strict(n_ext);
    for (size_t p0 = 0; p0 < n_polys; p0 += blk) {
        size_t nb = n_polys - p0 < blk ? n_polys - p0 : blk;
#pragma omp parallel for schedule(dynamic, 1) num_threads(nt)
        for (size_t q = 0; q < nb; q++) coset_lde_with_table(coeffs + (p0 + q) * n, n, rate_bits, shift, tmp + q * n_ext, &rt);
        /* transpose [P][n_ext] -> [n_ext][P], rows permuted by bit reversal */
#pragma omp parallel for schedule(static) num_threads(nt)
        for (size_t i = 0; i < n_ext; i++) {
            size_t src = glo_reverse_bits(i, lg);
            for (size_t q = 0; q < nb; q++) lv[i * n_polys + p0 + q] = tmp[q * n_ext + src];
        }
    }
    root_table_free(&rt);
    free(tmp);
    int rc = 0;
    if (digests && cap) rc = glo_merkle_tree(lv, n_ext, n_polys, cap_height, digests, cap, n_threads);
    if (!leaves) free(lv);
    return rc;
}

/* fri/oracle.rs:709-731 from_values: per-column ifft, then from_coeffs */
int glo_commit_from_values(const uint64_t *values, size_t n_polys, size_t n, unsigned rate_bits,
                           unsigned cap_height, uint64_t *coeffs, uint64_t *leaves, uint64_t *digests,
                           uint64_t *cap, int n_threads) {
    uint64_t *cf = coeffs ? coeffs : (uint64_t *)malloc(n_polys * n * 8);
    if (!cf) return -2;
    memcpy(cf, values, n_polys * n * 8);
    glo_fft_batch(cf, n_polys, n, 1, n_threads);
    int rc = glo_commit_from_coeffs(cf, n_polys, n, rate_bits, cap_height, leaves, digests, cap, n_threads);
    if (!coeffs) free(cf);
    return rc;
}

int glo_hardware_threads(void) {
#ifdef _OPENMP
    return omp_get_num_procs();
#else
    return 1;
#endif
}
