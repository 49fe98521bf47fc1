// poseidon_vector.h — the Poseidon permutation on the vector ALU alone: the MDS layers as 32-bit-half multiply-adds, the partial
// rounds in blocks of eleven with carry-free three-limb dot products. This was the product's permutation through round 2
// (2.24 G permutations/s); poseidon.h now runs the MDS layers of the full rounds on the matrix cores. Kept as the second,
// independent implementation: the diagnostic build serves gl_poseidon_permute_batch from it under PLONKY2_POSEIDON=vector
// (tests/test_gpu_merkle.py compares the two), tools/experiments/mds_mfma.hip times one against the other.
//
// Same permutation as Poseidon::poseidon (plonky2/src/hash/poseidon.rs:602-616) with the "fast"
// partial rounds (poseidon.rs:312-365, 400-427), restructured in blocks of eleven rounds (see
// partial_rounds). One thread owns one permutation; the 12-word state lives in 24 VGPRs for the whole
// permutation, inner loops are unrolled so table indices are uniform and the constants arrive as
// s_load / literals, not per-lane loads.
#pragma once
#include "gl_field.h"

#define POSEIDON_CONST __device__ const
#include "poseidon_constants.h"
#include "poseidon_limb_constants.h"

namespace poseidon_vector {

constexpr int W = 12;
constexpr int HALF_FULL = 4;
constexpr int N_PARTIAL = 22;

// MDS layer, state' = (circ(C) + diag(D)) * state  (poseidon.rs:174-194, 238-260), plus the NEXT
// layer's additive constants `rc_next` (constant_layer of the following round, poseidon.rs:484-493, or
// partial_first_constant_layer, :312-320): they ride on the accumulators' initial values for free.
// All C[i] <= 41 and D[0] = 8, so each row is accumulated exactly in two u64 lanes (low and high
// 32-bit halves of the state words; 2^32 + 12*41*2^32 < 2^42) and reduced once.
__device__ __forceinline__ void mds_layer(uint64_t (&s)[W], const uint64_t *__restrict__ rc_next) {
    uint64_t lo[W], hi[W];
#pragma unroll
    for (int i = 0; i < W; i++) {
        lo[i] = s[i] & 0xFFFFFFFFull;
        hi[i] = s[i] >> 32;
    }
#pragma unroll
    for (int r = 0; r < W; r++) {
        const uint64_t rc = rc_next[r];
        uint64_t al = rc & 0xFFFFFFFFull, ah = rc >> 32;
#pragma unroll
        for (int i = 0; i < W; i++) {
            al += lo[(i + r) % W] * POSEIDON_MDS_CIRC[i];
            ah += hi[(i + r) % W] * POSEIDON_MDS_CIRC[i];
        }
        al += lo[r] * POSEIDON_MDS_DIAG[r];
        ah += hi[r] * POSEIDON_MDS_DIAG[r];
        s[r] = gl::fold96(al, ah);  // al + ah*2^32 (< 2^75) mod p
    }
}

__device__ const uint64_t POSEIDON_ZERO_ROW[W] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

// s-box layer + MDS layer; the round's own constants were added by the previous layer
__device__ __forceinline__ void full_round(uint64_t (&s)[W], const uint64_t *__restrict__ rc_next) {
#pragma unroll
    for (int i = 0; i < W; i++) s[i] = gl::pow7(s[i]);
    mds_layer(s, rc_next);
}

// lazy dot products (one reduction per sum): gl::DotAcc / dot_term / dot_finish in gl_field.h
using gl::DotAcc;
using gl::dot_finish;
using gl::dot_term;

// The 22 partial rounds (poseidon.rs:400-427, 587-599; partial_first_constant_layer has already been
// added by the preceding MDS layer). The reference's recurrence per round is
//     u_r = sbox(s0) + rc_r;   d_r = c*u_r + sum_i s_i*w_hat[r][i];   s_i += u_r*v[r][i];   s0 = d_r
// i.e. eleven multiply-REDUCE-adds per round just to keep the s_i current. The s_i are linear in the
// u_q, so inside a block of eleven rounds they are left at their block-start values and the missing
// part is added through precomputed cross terms,
//     d_r = c*u_r + sum_i s_i(block start)*w_hat[r][i] + sum_{q in block, q < r} CROSS[r][q]*u_q,
//     CROSS[r][q] = sum_i w_hat[r][i]*v[q][i]      (tools/gen_poseidon_constants.py)
// and the s_i are brought up to date once per block, s_i += sum_q v[q][i]*u_q — all of it lazy dot
// products instead of 22-instruction macs: 638 terms and 44 reductions for the 22 rounds instead of 264 terms, 242 macs
// and 22 reductions — and, since round 2, carry-free ones: the multiplicand is split once into 21/21/22-bit limbs and every
// constant comes with its 2^21 and 2^42 multiples, six multiply-adds per term and a seven-instruction reduction per sum
// (gl::dot_term3 / fold96; before: four multiply-adds + four carry counters per term and twenty instructions per sum).
__device__ __forceinline__ void partial_rounds(uint64_t (&s)[W]) {
    using gl::DotAcc2;
    using gl::Limbs3;
    using gl::dot_term3;
    // mds_partial_layer_init (poseidon.rs:339-365): out[c] = sum_r s[r] * M[r-1][c-1]; constants in consumption order
    {
        Limbs3 sl[W];
#pragma unroll
        for (int r = 1; r < W; r++) sl[r] = gl::split21(s[r]);
#pragma unroll
        for (int c = 1; c < W; c++) {
            DotAcc2 acc;
#pragma unroll
            for (int r = 1; r < W; r++) {
                const int t = (c - 1) * 11 + (r - 1);
                dot_term3(acc, sl[r], POSEIDON_INIT_STREAM_C0[t], POSEIDON_INIT_STREAM_C21[t], POSEIDON_INIT_STREAM_C42[t]);
            }
            s[c] = gl::fold96(acc.lo, acc.hi);
        }
    }
    constexpr int B = 11;
#pragma unroll 1
    for (int blk = 0; blk < N_PARTIAL / B; blk++) {
        // this block's constants: [round k: 11 W_HATS, k CROSS] x 11, then VS [i][q] (tools/gen_poseidon_limb_tables.py)
        const uint64_t *__restrict__ c0 = POSEIDON_PARTIAL_STREAM_C0 + blk * POSEIDON_PARTIAL_BLOCK_TERMS;
        const uint64_t *__restrict__ c21 = POSEIDON_PARTIAL_STREAM_C21 + blk * POSEIDON_PARTIAL_BLOCK_TERMS;
        const uint64_t *__restrict__ c42 = POSEIDON_PARTIAL_STREAM_C42 + blk * POSEIDON_PARTIAL_BLOCK_TERMS;
        Limbs3 sl[W], ul[B];
#pragma unroll
        for (int i = 1; i < W; i++) sl[i] = gl::split21(s[i]);  // block-start values, used by all eleven rounds of the block
        uint64_t x = s[0];
#pragma unroll
        for (int k = 0; k < B; k++) {
            const uint64_t u = gl::add_canonical(gl::pow7(x), POSEIDON_FAST_PARTIAL_ROUND_CONSTANTS[blk * B + k]);
            ul[k] = gl::split21(u);
            DotAcc2 acc;
            gl::dot_term_small(acc, u, (uint32_t)(POSEIDON_MDS_CIRC[0] + POSEIDON_MDS_DIAG[0]));
            const int off = k * 11 + k * (k - 1) / 2;
#pragma unroll
            for (int i = 1; i < W; i++) dot_term3(acc, sl[i], c0[off + i - 1], c21[off + i - 1], c42[off + i - 1]);
#pragma unroll
            for (int q = 0; q < k; q++) dot_term3(acc, ul[q], c0[off + 11 + q], c21[off + 11 + q], c42[off + 11 + q]);
            x = gl::fold96(acc.lo, acc.hi);
        }
        s[0] = x;
#pragma unroll
        for (int i = 1; i < W; i++) {
            DotAcc2 acc;
            acc.lo = (uint32_t)s[i], acc.hi = s[i] >> 32;  // the block-start value, weight 2^0
#pragma unroll
            for (int q = 0; q < B; q++) {
                const int t = 176 + (i - 1) * 11 + q;
                dot_term3(acc, ul[q], c0[t], c21[t], c42[t]);
            }
            s[i] = gl::fold96(acc.lo, acc.hi);
        }
    }
}

__device__ __forceinline__ void permute(uint64_t (&s)[W]) {
    // constant_layer of round 0; every later constant layer is folded into the MDS layer before it
#pragma unroll
    for (int i = 0; i < W; i++) s[i] = gl::add_canonical(s[i], POSEIDON_ALL_ROUND_CONSTANTS[i]);
#pragma unroll 1
    for (int r = 0; r < HALF_FULL; r++)
        full_round(s, r + 1 < HALF_FULL ? POSEIDON_ALL_ROUND_CONSTANTS + W * (r + 1) : POSEIDON_FAST_PARTIAL_FIRST_ROUND_CONSTANT);
    partial_rounds(s);
#pragma unroll
    for (int i = 0; i < W; i++) s[i] = gl::add_canonical(s[i], POSEIDON_ALL_ROUND_CONSTANTS[i + W * (HALF_FULL + N_PARTIAL)]);
#pragma unroll 1
    for (int r = 0; r < HALF_FULL; r++)
        full_round(s, r + 1 < HALF_FULL ? POSEIDON_ALL_ROUND_CONSTANTS + W * (HALF_FULL + N_PARTIAL + r + 1) : POSEIDON_ZERO_ROW);
}

}  // namespace poseidon_vector
