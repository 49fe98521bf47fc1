// poseidon.cuh — Poseidon permutation over Goldilocks (width 12, 4+22+4 rounds, x^7) for gfx950.
//
// Same permutation as Poseidon::poseidon (plonky2/src/hash/poseidon.rs:602-616) with the "fast"
// partial rounds (poseidon.rs:312-365, 400-427). One thread owns one permutation; the 12-word
// state lives in 24 VGPRs for the whole permutation, every loop is unrolled so all table indices
// are compile-time and the constants arrive as scalar literals / s_load, not per-lane loads.
// Integer modular arithmetic only — no MFMA use is possible or attempted.
#pragma once
#include "gl_field.cuh"

#define POSEIDON_CONST __device__ const
#include "poseidon_constants.h"

namespace poseidon {

constexpr int W = 12;
constexpr int HALF_FULL = 4;
constexpr int N_PARTIAL = 22;

// MDS layer, state' = (circ(C) + diag(D)) * state  (poseidon.rs:174-194, 238-260).
// All C[i] <= 41 and D[0] = 8, so each row is accumulated exactly in two u64 lanes (low and
// high 32-bit halves of the state words; 12*41*2^32 < 2^42) and reduced once.
__device__ __forceinline__ void mds_layer(uint64_t (&s)[W]) {
    uint64_t lo[W], hi[W];
#pragma unroll
    for (int i = 0; i < W; i++) {
        lo[i] = s[i] & 0xFFFFFFFFull;
        hi[i] = s[i] >> 32;
    }
#pragma unroll
    for (int r = 0; r < W; r++) {
        uint64_t al = 0, ah = 0;
#pragma unroll
        for (int i = 0; i < W; i++) {
            al += lo[(i + r) % W] * POSEIDON_MDS_CIRC[i];
            ah += hi[(i + r) % W] * POSEIDON_MDS_CIRC[i];
        }
        al += lo[r] * POSEIDON_MDS_DIAG[r];
        ah += hi[r] * POSEIDON_MDS_DIAG[r];
        // value = al + ah*2^32  (< 2^75): fold into (lo64, hi32)
        uint64_t l = al + (ah << 32);
        uint32_t h = (uint32_t)(ah >> 32) + (l < al ? 1u : 0u);
        s[r] = gl::reduce96(l, h);
    }
}

__device__ __forceinline__ void full_round(uint64_t (&s)[W], int round_ctr) {
#pragma unroll
    for (int i = 0; i < W; i++) s[i] = gl::add_canonical(s[i], POSEIDON_ALL_ROUND_CONSTANTS[i + W * round_ctr]);
#pragma unroll
    for (int i = 0; i < W; i++) s[i] = gl::pow7(s[i]);
    mds_layer(s);
}

// ---- lazy dot products --------------------------------------------------------------------
// sum_i a_i * b_i over Goldilocks with ONE reduction at the end (the reference does the same on
// the CPU with a u160 accumulator, poseidon.rs:34-47, 400-413). A 64x64 product is four 32x32
// partial products; they are accumulated column-wise,
//     A0 += a0*b0        A1 += a0*b1 + a1*b0        A2 += a1*b1
// each column in a wrapping 64-bit accumulator (v_mad_u64_u32 adds for free) plus a 32-bit count
// of its wrap-arounds (the mad's carry-out, one v_addc each). 8 half-rate instructions per term
// instead of ~20 for a multiply-reduce-add. The carry SGPRs are consumed >= 2 instructions
// after they are written (gfx950 VALU-writes-SGPR -> VALU-reads-it hazard).
struct DotAcc {
    uint64_t a0 = 0, a1 = 0, a2 = 0;
    uint32_t k0 = 0, k1 = 0, k2 = 0;
};

__device__ __forceinline__ void dot_term(DotAcc &d, uint64_t a, uint64_t b) {
    uint32_t al = (uint32_t)a, ah = (uint32_t)(a >> 32);
    uint32_t bl = (uint32_t)b, bh = (uint32_t)(b >> 32);
    uint64_t c0, c1, c2;
    asm("v_mad_u64_u32 %0, %6, %9, %11, %0\n\t"
        "v_mad_u64_u32 %1, %7, %9, %12, %1\n\t"
        "v_mad_u64_u32 %2, %8, %10, %12, %2\n\t"
        "v_addc_co_u32_e64 %3, vcc, 0, %3, %6\n\t"
        "v_mad_u64_u32 %1, %6, %10, %11, %1\n\t"
        "v_addc_co_u32_e64 %4, vcc, 0, %4, %7\n\t"
        "v_addc_co_u32_e64 %5, vcc, 0, %5, %8\n\t"
        "v_addc_co_u32_e64 %4, vcc, 0, %4, %6"
        : "+v"(d.a0), "+v"(d.a1), "+v"(d.a2), "+v"(d.k0), "+v"(d.k1), "+v"(d.k2), "=&s"(c0), "=&s"(c1), "=&s"(c2)
        : "v"(al), "v"(ah), "s"(bl), "s"(bh)
        : "vcc");
}

// value = (A0 + k0*2^64) + (A1 + k1*2^64)*2^32 + (A2 + k2*2^64)*2^64   (mod p)
__device__ __forceinline__ uint64_t dot_finish(const DotAcc &d) {
    uint64_t lo = d.a0 + (d.a1 << 32);
    uint64_t c0 = lo < d.a0;
    uint64_t h1 = (d.a1 >> 32) + c0 + d.k0;  // < 2^33, no wrap
    uint64_t hi = h1 + d.a2;
    uint64_t top = (uint64_t)(hi < h1) + d.k2;  // units of 2^128
    uint64_t k1s = (uint64_t)d.k1 << 32;  // k1 * 2^96 = (k1 << 32) * 2^64
    uint64_t hi2 = hi + k1s;
    top += hi2 < k1s;
    // 2^128 = -2^32 (mod p); top < 2^7 so top << 32 is canonical
    return gl::sub(gl::reduce128(lo, hi2), top << 32);
}

__device__ __forceinline__ void partial_rounds(uint64_t (&s)[W]) {
    // partial_first_constant_layer (poseidon.rs:312-320)
#pragma unroll
    for (int i = 0; i < W; i++) s[i] = gl::add_canonical(s[i], POSEIDON_FAST_PARTIAL_FIRST_ROUND_CONSTANT[i]);
    // mds_partial_layer_init (poseidon.rs:339-365): out[c] = sum_r s[r] * M[r-1][c-1]
    {
        uint64_t out[W];
        out[0] = s[0];
#pragma unroll
        for (int c = 1; c < W; c++) {
            DotAcc acc;
#pragma unroll
            for (int r = 1; r < W; r++) dot_term(acc, s[r], POSEIDON_FAST_PARTIAL_ROUND_INITIAL_MATRIX[(r - 1) * 11 + (c - 1)]);
            out[c] = dot_finish(acc);
        }
#pragma unroll
        for (int i = 0; i < W; i++) s[i] = out[i];
    }
#pragma unroll 1
    for (int r = 0; r < N_PARTIAL; r++) {
        s[0] = gl::pow7(s[0]);
        s[0] = gl::add_canonical(s[0], POSEIDON_FAST_PARTIAL_ROUND_CONSTANTS[r]);
        // mds_partial_layer_fast (poseidon.rs:400-427): d = s0*(C0+D0) + sum_i s[i]*w_hat[i]
        DotAcc acc;
        dot_term(acc, s[0], POSEIDON_MDS_CIRC[0] + POSEIDON_MDS_DIAG[0]);
#pragma unroll
        for (int i = 1; i < W; i++) dot_term(acc, s[i], POSEIDON_FAST_PARTIAL_ROUND_W_HATS[r * 11 + (i - 1)]);
        uint64_t s0 = s[0];
#pragma unroll
        for (int i = 1; i < W; i++) s[i] = gl::mac(s[i], s0, POSEIDON_FAST_PARTIAL_ROUND_VS[r * 11 + (i - 1)]);
        s[0] = dot_finish(acc);
    }
}

__device__ __forceinline__ void permute(uint64_t (&s)[W]) {
#pragma unroll 1
    for (int r = 0; r < HALF_FULL; r++) full_round(s, r);
    partial_rounds(s);
#pragma unroll 1
    for (int r = 0; r < HALF_FULL; r++) full_round(s, HALF_FULL + N_PARTIAL + r);
}

}  // namespace poseidon
