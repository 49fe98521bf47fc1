// ubench_field.hip — candidate formulations of gl::add / gl::sub / gl::mul that trade carry-flag instructions
// (5 cycles each on gfx950, tools/ubench_issue.hip) for multiply-adds, a flag-free 64-bit add and a wave-uniform
// branch around the second wrap correction. Each candidate is checked against the library's primitive on random,
// edge and deliberately non-canonical operands (the rare path), then timed: 4 dependent chains per thread, 8 waves/SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I plonky2_gpu_amd/csrc tools/ubench_field.hip -o tools/ubench_field
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#include "gl_field.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

namespace cand {

// a + b, any representatives. s = a + b (carry c); r = s + (c ? eps : 0) as ONE multiply-add whose carry-out tells
// whether the second correction is needed — only possible when a, b >= p, so it sits behind a wave-uniform branch.
__device__ __forceinline__ uint64_t add(uint64_t a, uint64_t b) {
    uint32_t al = (uint32_t)a, ah = (uint32_t)(a >> 32), bl = (uint32_t)b, bh = (uint32_t)(b >> 32);
    uint64_t r;
    asm("v_add_co_u32_e32 v116, vcc, %1, %3\n\t"
        "v_addc_co_u32_e32 v117, vcc, %2, %4, vcc\n\t"
        "v_cndmask_b32_e64 v126, 0, -1, vcc\n\t"                  // overflow: 2^64 = 2^32 - 1
        "v_mad_u64_u32 %0, vcc, v126, 1, v[116:117]\n\t"          // r = s + t, carry -> vcc
        "s_cbranch_vccz 1f\n\t"
        "v_cndmask_b32_e64 v126, 0, -1, vcc\n\t"                  // rare: once more (cannot overflow a third time)
        "v_mad_u64_u32 %0, vcc, v126, 1, %0\n\t"
        "1:"
        : "=&v"(r)
        : "v"(al), "v"(ah), "v"(bl), "v"(bh)
        : "vcc", "v116", "v117", "v126");
    return r;
}

__device__ __forceinline__ uint64_t sub(uint64_t a, uint64_t b) {
    uint32_t al = (uint32_t)a, ah = (uint32_t)(a >> 32), bl = (uint32_t)b, bh = (uint32_t)(b >> 32);
    uint32_t rl, rh, t;
    asm("v_sub_co_u32_e32 %0, vcc, %3, %5\n\t"
        "v_subb_co_u32_e32 %1, vcc, %4, %6, vcc\n\t"
        "v_cndmask_b32_e64 %2, 0, -1, vcc\n\t"                    // borrow: -2^64 = -(2^32 - 1)
        "v_sub_co_u32_e32 %0, vcc, %0, %2\n\t"
        "v_subbrev_co_u32_e32 %1, vcc, 0, %1, vcc\n\t"
        "s_cbranch_vccz 1f\n\t"
        "v_cndmask_b32_e64 %2, 0, -1, vcc\n\t"                    // rare: once more
        "v_sub_co_u32_e32 %0, vcc, %0, %2\n\t"
        "v_subbrev_co_u32_e32 %1, vcc, 0, %1, vcc\n\t"
        "1:"
        : "=&v"(rl), "=&v"(rh), "=&v"(t)
        : "v"(al), "v"(ah), "v"(bl), "v"(bh)
        : "vcc");
    return ((uint64_t)rh << 32) | rl;
}

// a * b. Product as in gl::mul. Reduction x = lo - (hh + c1) + hl*eps as in gl::mul, but: the borrow correction
// (lo < hh + c1: probability ~2^-32 on real data) sits behind a wave-uniform branch, and the carry correction after
// r = t0 + hl*eps is ONE multiply-add (t*1 + r) that writes the result pair.
__device__ __forceinline__ uint64_t mul(uint64_t a, uint64_t b) {
    uint32_t al = (uint32_t)a, ah = (uint32_t)(a >> 32), bl = (uint32_t)b, bh = (uint32_t)(b >> 32);
    uint64_t r, c1;
    asm("v_mad_u64_u32 v[116:117], vcc, %2, %4, 0\n\t"            // T = al*bl
        "v_mad_u64_u32 v[118:119], vcc, %2, %5, 0\n\t"            // U = al*bh
        "v_mad_u64_u32 v[120:121], %1, %3, %4, v[118:119]\n\t"    // V = ah*bl + U, carry c1 (weight 2^96)
        "v_mad_u64_u32 v[122:123], vcc, %3, %5, 0\n\t"            // W = ah*bh
        "v_add_co_u32_e32 v117, vcc, v117, v120\n\t"              // lo = (v116, v117)
        "v_addc_co_u32_e32 v122, vcc, v122, v121, vcc\n\t"        // hl
        "v_addc_co_u32_e32 v123, vcc, 0, v123, vcc\n\t"           // hh (without c1)
        "v_subb_co_u32_e64 v116, vcc, v116, v123, %1\n\t"         // t0 = lo - hh - c1
        "v_subbrev_co_u32_e32 v117, vcc, 0, v117, vcc\n\t"
        "s_cbranch_vccz 1f\n\t"
        "v_cndmask_b32_e64 v126, 0, -1, vcc\n\t"                  // rare: borrow => t0 -= eps (cannot borrow again)
        "v_sub_co_u32_e32 v116, vcc, v116, v126\n\t"
        "v_subbrev_co_u32_e32 v117, vcc, 0, v117, vcc\n\t"
        "1:\n\t"
        "v_mad_u64_u32 v[116:117], vcc, v122, -1, v[116:117]\n\t" // r = t0 + hl*eps, carry -> vcc
        "v_cndmask_b32_e64 v126, 0, -1, vcc\n\t"                  // carry: += eps (cannot carry again)
        "v_mad_u64_u32 %0, vcc, v126, 1, v[116:117]"
        : "=&v"(r), "=&s"(c1)
        : "v"(al), "v"(ah), "v"(bl), "v"(bh)
        : "vcc", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v126");
    return r;
}

}  // namespace cand

template <int OP>
__device__ __forceinline__ uint64_t apply(uint64_t a, uint64_t b) {
    if constexpr (OP == 0) return gl::add(a, b);
    if constexpr (OP == 1) return cand::add(a, b);
    if constexpr (OP == 2) return gl::sub(a, b);
    if constexpr (OP == 3) return cand::sub(a, b);
    if constexpr (OP == 4) return gl::mul(a, b);
    if constexpr (OP == 5) return cand::mul(a, b);
    return a;
}

template <int OP>
__global__ void check_kernel(const uint64_t *a, const uint64_t *b, uint64_t *out, uint64_t n) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = gl::canon(apply<OP>(a[i], b[i]));
}

template <int OP>
__global__ __launch_bounds__(256) void time_kernel(uint64_t *out, uint64_t seed, int iters) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t a[4], b = seed ^ 0xD1B54A32D192ED03ull;
#pragma unroll
    for (int k = 0; k < 4; k++) a[k] = seed * (i + 1) + k * 0x9E3779B97F4A7C15ull;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int k = 0; k < 4; k++) a[k] = apply<OP>(a[k], b);
    }
    out[i] = a[0] ^ a[1] ^ a[2] ^ a[3];
}

static uint64_t rnd(uint64_t &s) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    return s;
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount, mhz = prop.clockRate / 1000;
    const uint64_t P = 0xFFFFFFFF00000001ull;
    // operands: edges x edges, the non-canonical window [p, 2^64) x itself (the second-correction path), random
    std::vector<uint64_t> edge = {0, 1, 2, 0xFFFFFFFFull, 0x100000000ull, 0x100000001ull, 0x7FFFFFFFFFFFFFFFull, 0x8000000000000000ull,
                                  P - 2, P - 1, P, P + 1, P + 2, P + 0x7FFFFFFFull, 0xFFFFFFFFFFFFFFFEull, 0xFFFFFFFFFFFFFFFFull,
                                  0xFFFFFFFF00000000ull, 0xFFFFFFFEFFFFFFFFull, 0x00000001FFFFFFFFull, 0xFFFFFFFF80000000ull};
    std::vector<uint64_t> ha, hb;
    for (uint64_t x : edge) for (uint64_t y : edge) { ha.push_back(x); hb.push_back(y); }
    for (int i = 0; i < 64; i++) for (int j = 0; j < 64; j++) {  // products with empty low halves: the borrow path of mul
        ha.push_back(1ull << i); hb.push_back(1ull << j);
        ha.push_back((1ull << i) | 1); hb.push_back(0xFFFFFFFFFFFFFFFFull << j);
    }
    uint64_t s = 88172645463325252ull;
    for (int i = 0; i < 200000; i++) { ha.push_back(P + rnd(s) % 0xFFFFFFFFull); hb.push_back(P + rnd(s) % 0xFFFFFFFFull); }  // both >= p
    for (int i = 0; i < 200000; i++) { ha.push_back(rnd(s) % 8); hb.push_back(P + rnd(s) % 0xFFFFFFFFull); }
    for (int i = 0; i < 200000; i++) { ha.push_back(rnd(s) & 0xFFFFFFFFull); hb.push_back(rnd(s) | 0xFFFFFFFF00000000ull); }
    for (int i = 0; i < 4000000; i++) { ha.push_back(rnd(s)); hb.push_back(rnd(s)); }
    const uint64_t n = ha.size();
    uint64_t *da, *db, *d0, *d1;
    CK(hipMalloc(&da, n * 8)); CK(hipMalloc(&db, n * 8)); CK(hipMalloc(&d0, n * 8)); CK(hipMalloc(&d1, n * 8));
    CK(hipMemcpy(da, ha.data(), n * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, hb.data(), n * 8, hipMemcpyHostToDevice));
    std::vector<uint64_t> r0(n), r1(n);
    const char *names[3] = {"add", "sub", "mul"};
    auto check = [&](int which, auto k0, auto k1) {
        hipLaunchKernelGGL(k0, dim3((n + 255) / 256), dim3(256), 0, 0, da, db, d0, n);
        hipLaunchKernelGGL(k1, dim3((n + 255) / 256), dim3(256), 0, 0, da, db, d1, n);
        // and with ONE wavefront alone on the device (back-to-back issue, the hazard-revealing case) on the first 64K
        CK(hipMemcpy(r0.data(), d0, n * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(r1.data(), d1, n * 8, hipMemcpyDeviceToHost));
        uint64_t bad = 0, big = 0;
        for (uint64_t i = 0; i < n; i++) {
            unsigned __int128 x = ha[i] % P, y = hb[i] % P;
            uint64_t ref = which == 0 ? (uint64_t)((x + y) % P) : which == 1 ? (uint64_t)((x + P - y) % P) : (uint64_t)(x * y % P);
            if (r0[i] != ref) big++;
            if (r1[i] != ref) { if (bad < 5) printf("  %s MISMATCH a=%016llx b=%016llx got %016llx want %016llx\n", names[which], (unsigned long long)ha[i], (unsigned long long)hb[i], (unsigned long long)r1[i], (unsigned long long)ref); bad++; }
        }
        printf("%s: %llu operand pairs, candidate mismatches %llu, library mismatches vs big-int %llu\n", names[which], (unsigned long long)n, (unsigned long long)bad, (unsigned long long)big);
        return bad == 0;
    };
    bool ok = check(0, check_kernel<0>, check_kernel<1>);
    ok &= check(1, check_kernel<2>, check_kernel<3>);
    ok &= check(2, check_kernel<4>, check_kernel<5>);
    // single-wave run of the candidates (one block of 64 threads): same answers?
    {
        const uint64_t m = 64 * 1024;
        auto single = [&](auto k, uint64_t *dst) {
            for (uint64_t off = 0; off < m; off += 64) hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da + off, db + off, dst + off, (uint64_t)64);
            CK(hipDeviceSynchronize());
        };
        std::vector<uint64_t> s1(m);
        uint64_t bad = 0;
        single(check_kernel<1>, d1); CK(hipMemcpy(s1.data(), d1, m * 8, hipMemcpyDeviceToHost));
        for (uint64_t i = 0; i < m; i++) { unsigned __int128 x = ha[i] % P, y = hb[i] % P; bad += s1[i] != (uint64_t)((x + y) % P); }
        single(check_kernel<3>, d1); CK(hipMemcpy(s1.data(), d1, m * 8, hipMemcpyDeviceToHost));
        for (uint64_t i = 0; i < m; i++) { unsigned __int128 x = ha[i] % P, y = hb[i] % P; bad += s1[i] != (uint64_t)((x + P - y) % P); }
        single(check_kernel<5>, d1); CK(hipMemcpy(s1.data(), d1, m * 8, hipMemcpyDeviceToHost));
        for (uint64_t i = 0; i < m; i++) { unsigned __int128 x = ha[i] % P, y = hb[i] % P; bad += s1[i] != (uint64_t)(x * y % P); }
        printf("single-wavefront launches (64K pairs x 3 ops): mismatches %llu\n", (unsigned long long)bad);
        ok &= bad == 0;
    }
    printf("device: %s, CUs %d, %d MHz\n", prop.name, cus, mhz);
    const int iters = 2048, blocks = cus * 8;
    uint64_t *out;
    CK(hipMalloc(&out, (size_t)blocks * 256 * 8));
    auto timeit = [&](const char *label, auto k) {
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        float best = 1e30f;
        for (int r = 0; r < 4; r++) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, 12345ull + r, iters);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (r && ms < best) best = ms;
        }
        printf("%-26s %8.3f ms  %6.1f cycles per op and wave (8 waves/SIMD)\n", label, best, best * 1e-3 * mhz * 1e6 / (8.0 * iters * 4));
    };
    timeit("gl::add   (library)", time_kernel<0>);
    timeit("cand::add", time_kernel<1>);
    timeit("gl::sub   (library)", time_kernel<2>);
    timeit("cand::sub", time_kernel<3>);
    timeit("gl::mul   (library)", time_kernel<4>);
    timeit("cand::mul", time_kernel<5>);
    printf(ok ? "ALL CANDIDATES AGREE\n" : "CANDIDATE MISMATCH\n");
    return ok ? 0 : 1;
}
