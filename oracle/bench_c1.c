/* bench_c1.c — BASELINE.json configs[0] ("plumbing, no GPU"): the reference's criterion workloads
 * plonky2/benches/field_arithmetic.rs:11-177 and plonky2/benches/ffts.rs:9-38 restated over the CPU
 * oracle (gl_oracle.c, a C restatement of the reference's generic scalar path — NOT the Rust binary).
 * TEST/BENCH INFRASTRUCTURE ONLY. Prints one JSON object: nanoseconds per iteration of each workload,
 * single thread, like criterion's per-iteration estimate. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "gl_oracle.h"

static uint64_t rng_state = 0x706C6F6E6B7932ull;
static uint64_t rnd(void) { /* SplitMix64 reduced by rejection */
    for (;;) {
        uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        if (z < 0xFFFFFFFF00000001ull) return z;
    }
}
static double now(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return t.tv_sec * 1e9 + t.tv_nsec;
}
static volatile uint64_t sink;

/* batch_multiplicative_inverse (field/src/types.rs:133-223): Montgomery's trick, one inversion */
static void batch_inverse(const uint64_t *x, uint64_t *out, size_t n) {
    uint64_t acc = 1;
    for (size_t i = 0; i < n; i++) {
        out[i] = acc;
        acc = glo_mul(acc, x[i]);
    }
    uint64_t inv = glo_inverse(acc);
    for (size_t i = n; i-- > 0;) {
        out[i] = glo_mul(out[i], inv);
        inv = glo_mul(inv, x[i]);
    }
}

#define TIME(name, reps, setup, ...)                        \
    do {                                                    \
        double best = 1e300;                                \
        for (int trial = 0; trial < 5; trial++) {           \
            double t0 = now();                              \
            for (long it = 0; it < (reps); it++) {          \
                setup;                                      \
                __VA_ARGS__;                                \
            }                                               \
            double dt = (now() - t0) / (reps);              \
            if (dt < best) best = dt;                       \
        }                                                   \
        printf("%s\"%s\": %.2f", first ? "" : ", ", name, best); \
        first = 0;                                          \
    } while (0)

int main(void) {
    int first = 1;
    printf("{");
    uint64_t x = rnd(), y = rnd(), z = rnd(), w = rnd();
    TIME("mul-throughput (4 chains x 25)", 200000, , for (int k = 0; k < 25; k++) {
        uint64_t a = glo_mul(x, y), b = glo_mul(y, z), c = glo_mul(z, w), d = glo_mul(w, x);
        x = a; y = b; z = c; w = d; } sink = x ^ y ^ z ^ w);
    TIME("mul-latency (100 dependent)", 100000, , for (int k = 0; k < 100; k++) x = glo_mul(x, x); sink = x);
    TIME("sqr-throughput (4 chains x 25)", 200000, , for (int k = 0; k < 25; k++) {
        x = glo_mul(x, x); y = glo_mul(y, y); z = glo_mul(z, z); w = glo_mul(w, w); } sink = x ^ y ^ z ^ w);
    uint64_t v[10];
    for (int i = 0; i < 10; i++) v[i] = rnd();
    TIME("add-throughput (10 chains x 10)", 500000, , for (int k = 0; k < 10; k++) {
        uint64_t t[10];
        for (int i = 0; i < 10; i++) t[i] = glo_add(v[i], v[(i + 1) % 10]);
        memcpy(v, t, sizeof t); } sink = v[0]);
    TIME("add-latency (100 dependent)", 200000, , for (int k = 0; k < 100; k++) x = glo_add(x, x); sink = x);
    TIME("try_inverse", 20000, x = rnd(), sink = glo_inverse(x));
    static uint64_t bx[65536], bo[65536];
    for (int i = 0; i < 65536; i++) bx[i] = rnd() | 1;
    TIME("batch_multiplicative_inverse-tiny (2)", 20000, bx[0] = rnd() | 1, batch_inverse(bx, bo, 2); sink = bo[1]);
    TIME("batch_multiplicative_inverse-small (4)", 20000, bx[0] = rnd() | 1, batch_inverse(bx, bo, 4); sink = bo[3]);
    TIME("batch_multiplicative_inverse-medium (16)", 20000, bx[0] = rnd() | 1, batch_inverse(bx, bo, 16); sink = bo[15]);
    TIME("batch_multiplicative_inverse-large (256)", 5000, bx[0] = rnd() | 1, batch_inverse(bx, bo, 256); sink = bo[255]);
    TIME("batch_multiplicative_inverse-huge (65536)", 20, bx[0] = rnd() | 1, batch_inverse(bx, bo, 65536); sink = bo[65535]);
    for (int lg = 13; lg <= 16; lg++) {
        size_t n = (size_t)1 << lg;
        uint64_t *c = malloc(n * 8), *t = malloc(n * 8);
        for (size_t i = 0; i < n; i++) c[i] = rnd();
        char name[64];
        snprintf(name, sizeof name, "fft/%zu", n);
        TIME(name, 200 >> (lg - 13), memcpy(t, c, n * 8), glo_fft(t, n, 0); sink = t[1]);
        /* lde: zero-pad orig -> n, then fft_with_options(Some(3)) (ffts.rs:22-38) */
        size_t orig = n >> 3;
        snprintf(name, sizeof name, "lde/%zu", n);
        TIME(name, 200 >> (lg - 13), (memset(t, 0, n * 8), memcpy(t, c, orig * 8)), glo_fft(t, n, 3); sink = t[1]);
        free(c);
        free(t);
    }
    printf("}\n");
    return 0;
}
