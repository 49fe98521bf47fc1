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

// lazy dot products (one reduction per sum): gl::DotAcc / dot_term / dot_finish in gl_field.cuh
using gl::DotAcc;
using gl::dot_finish;
using gl::dot_term;

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
