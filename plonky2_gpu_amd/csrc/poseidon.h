// poseidon.h — Poseidon permutation over Goldilocks (width 12, 4+22+4 rounds, x^7) for gfx950.
//
// Same permutation as Poseidon::poseidon (plonky2/src/hash/poseidon.rs:602-616), computed the way `poseidon_naive`
// (poseidon.rs:565-585) states it: thirty rounds of constant layer, s-box layer (all twelve elements in rounds 0-3 and 26-29,
// element 0 in the 22 rounds between) and MDS layer (poseidon.rs:174-194, 238-260, 484-493). One thread owns one permutation;
// the 12-word state lives in 24 VGPRs throughout.
//
// The MDS layer runs on the MATRIX CORES — as an integer product, the only use this code base has for them.
// state' = (circ(C) + diag(D)) state with C[i] <= 41: split every state word into its eight bytes and the layer is eight
// independent 12 x 12 products of small integers, one per byte plane, recombined with weights 2^(8b):
//     plane_b[r] = sum_j M[r][j] * byte_b(s_j)  <=  256 * 255 < 2^16        (every row of circ(C) sums to 256)
// `v_mfma_i32_32x32x32_i8` multiplies a 32 x 32 by a 32 x 32 matrix of signed bytes; lane l supplies sixteen k — half l >> 5 of
// the 32 — of column l & 31 and receives sixteen rows of that column, rows (q & 3) + 8 (q >> 2) + 4 (l >> 5). The two lanes
// that share a column therefore feed DIFFERENT halves of k and read DIFFERENT rows: with an A operand whose rows are non-zero only
// in the half of k of the lanes that will read them, each lane's sixteen bytes (twelve state words + four unused) meet the
// twelve rows of the MDS matrix and the twelve results come back to the very same lane. No state moves between lanes; one
// instruction per byte plane, eight per layer, on a pipe the vector ALU does not wait for. (The first version used three
// `v_mfma_i32_16x16x64_i8` per plane, four rows each: 2.53 against 2.76 G permutations/s, profiles/r03_poseidon_matrix_cores.jsonl.)
// What the vector ALU still does per layer:
//   * 4 x 4 byte transpositions (v_perm_b32) so that one dword holds the same byte of four state words, and ^ 0x80 because
//     the matrix cores read SIGNED bytes (byte - 128; the accumulator input C = 128 * 256 puts the offset back),
//   * per output word: pack the eight 16-bit plane sums into four dwords (planes 0|2, 1|3, 4|6, 5|7), two multiply-adds
//     (x 256) to make the two 64-bit column sums of gl::fold96, and the reduction. The high dword of each sum's register pair
//     has to be initialised anyway: the NEXT layer's additive constant rides there as (X, Y) with
//     X 2^32 + Y 2^64 = c (mod p)  (tools/gen_poseidon_limb_tables.py solve_xy),
//   * the one diagonal entry (8 * s_0), two multiply-adds.
// 242 vector instructions + 8 matrix instructions per layer instead of 288 multiply-adds + 100: 34 against 54 us per layer and 2^22 states
// (tools/experiments/mds_mfma.hip, profiles/r03_poseidon_matrix_cores.jsonl).
//
// The matrix instruction reads and writes ALL 64 lanes' registers whatever EXEC says, and the A operand held by lane l is a row of the
// matrix that 32 OTHER lanes' results depend on: a kernel must build MdsOperands, and keep calling permute, with every lane of the wave active (clamp
// indices and predicate the stores instead of returning early). permute() traps when that is not so.
#pragma once
#include "poseidon_vector.h"

namespace poseidon {

using poseidon_vector::HALF_FULL;
using poseidon_vector::N_PARTIAL;
using poseidon_vector::W;

typedef int v4i32 __attribute__((ext_vector_type(4)));
typedef int v16i32 __attribute__((ext_vector_type(16)));

#ifdef POSEIDON_MDS_NATURAL
#include POSEIDON_MDS_NATURAL  // tools/experiments/mds_natural.h: the A/B build of round 6 (state fed as it lies, 18 matrix instructions per layer)
#else
struct MdsOperands {
    v4i32 A;   // this lane's row of the A operand: a row of the MDS matrix in its own half's sixteen k, or zero
    v16i32 C;  // 128 * (row sum) in every element
};

// POSEIDON_MDS_ROW_WORDS[o] for a per-lane o < 12 as a chain of selects on literals (no table in memory, no branches)
__device__ __forceinline__ uint32_t mds_row_word(uint32_t o) {
    constexpr uint32_t WORDS[12] = POSEIDON_MDS_ROW_WORDS;
    uint32_t x = 0;
#pragma unroll
    for (int k = 0; k < 12; k++) x = o == (uint32_t)k ? WORDS[k] : x;
    return x;
}

// Pure function of the lane number; call it before anything diverges.
__device__ __forceinline__ MdsOperands mds_operands() {
    const uint32_t lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const uint32_t m = lane & 31, h = lane >> 5;
    // Result register q of lane (n, h) is row (q & 3) + 8 (q >> 2) + 4 h of the product, and q is to be MDS row q: so row m of A
    // carries MDS row (m & 3) + 4 (m >> 3) — if that is below 12 — in the k of half (m >> 2) & 1, zeros in the other half.
    const uint32_t r = (m & 3) + 4 * (m >> 3);
    const bool on = ((m >> 2) & 1) == h && r < 12;
    MdsOperands o;
#pragma unroll
    for (int w = 0; w < 3; w++) o.A[w] = on ? (int)mds_row_word((4 * w + 12 - r) % 12) : 0;  // bytes CIRC[(4w + t - r) mod 12], t = 0..3
    o.A[3] = 0;  // k = 12..15 of either half: the fourth dword of a B operand may hold anything
#pragma unroll
    for (int k = 0; k < 16; k++) o.C[k] = POSEIDON_MDS_PLANE_OFFSET;
    // keep the operands in registers of their own for the whole kernel instead of re-deriving them per use
    asm volatile("" : "+v"(o.A), "+v"(o.C));
    return o;
}

__device__ __forceinline__ void require_full_wave() {
    if (__builtin_amdgcn_read_exec() != ~0ull) __builtin_trap();
}

#ifdef POSEIDON_MDS_LAYER
#include POSEIDON_MDS_LAYER  // an experiment's mds_layer in place of the one below (tools/experiments/mds_interleave.h)
#else
// MDS layer + the additive constants of whatever follows, xy = [12][X, Y] (poseidon_limb_constants.h).
__device__ __forceinline__ void mds_layer(uint64_t (&s)[W], const MdsOperands &ops, const uint32_t *__restrict__ xy) {
    v4i32 T[8];  // T[b] = byte b of words 0-3 | 4-7 | 8-11 | (never written: meets zero columns of A)
#pragma unroll
    for (int G = 0; G < 3; G++)
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const uint32_t r0 = (uint32_t)(s[4 * G] >> (32 * h)), r1 = (uint32_t)(s[4 * G + 1] >> (32 * h));
            const uint32_t r2 = (uint32_t)(s[4 * G + 2] >> (32 * h)), r3 = (uint32_t)(s[4 * G + 3] >> (32 * h));
            // v_perm_b32 D, S0, S1, sel: bytes 0-3 of the selector space are S1's, 4-7 S0's
            const uint32_t a01 = __builtin_amdgcn_perm(r1, r0, 0x05010400u), c01 = __builtin_amdgcn_perm(r1, r0, 0x07030602u);
            const uint32_t a23 = __builtin_amdgcn_perm(r3, r2, 0x05010400u), c23 = __builtin_amdgcn_perm(r3, r2, 0x07030602u);
            T[4 * h + 0][G] = (int)(__builtin_amdgcn_perm(a23, a01, 0x05040100u) ^ 0x80808080u);
            T[4 * h + 1][G] = (int)(__builtin_amdgcn_perm(a23, a01, 0x07060302u) ^ 0x80808080u);
            T[4 * h + 2][G] = (int)(__builtin_amdgcn_perm(c23, c01, 0x05040100u) ^ 0x80808080u);
            T[4 * h + 3][G] = (int)(__builtin_amdgcn_perm(c23, c01, 0x07060302u) ^ 0x80808080u);
        }
    const uint32_t x0l = (uint32_t)s[0], x0h = (uint32_t)(s[0] >> 32);
    uint64_t al[W], ah[W];
    // The packing of the plane sums is left to the compiler: it knows the wait states between a matrix instruction and the first
    // vector instruction that reads its result; inline asm is opaque to that.
    {
        const v16i32 D0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(ops.A, T[0], ops.C, 0, 0, 0), D2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(ops.A, T[2], ops.C, 0, 0, 0);
        const v16i32 D1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(ops.A, T[1], ops.C, 0, 0, 0), D3 = __builtin_amdgcn_mfma_i32_32x32x32_i8(ops.A, T[3], ops.C, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < W; r++) {
            const uint32_t Al = (uint32_t)D0[r] | ((uint32_t)D2[r] << 16), Bl = (uint32_t)D1[r] | ((uint32_t)D3[r] << 16);
            al[r] = ((uint64_t)xy[2 * r] << 32) | Al;
            asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(al[r]) : "v"(Bl), "s"(256u) : "vcc");
        }
    }
    {
        const v16i32 D4 = __builtin_amdgcn_mfma_i32_32x32x32_i8(ops.A, T[4], ops.C, 0, 0, 0), D6 = __builtin_amdgcn_mfma_i32_32x32x32_i8(ops.A, T[6], ops.C, 0, 0, 0);
        const v16i32 D5 = __builtin_amdgcn_mfma_i32_32x32x32_i8(ops.A, T[5], ops.C, 0, 0, 0), D7 = __builtin_amdgcn_mfma_i32_32x32x32_i8(ops.A, T[7], ops.C, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < W; r++) {
            const uint32_t Ah = (uint32_t)D4[r] | ((uint32_t)D6[r] << 16), Bh = (uint32_t)D5[r] | ((uint32_t)D7[r] << 16);
            ah[r] = ((uint64_t)xy[2 * r + 1] << 32) | Ah;
            asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(ah[r]) : "v"(Bh), "s"(256u) : "vcc");
        }
    }
    // the diagonal entry stays out of the matrix product (it would push a plane's sum past 16 bits)
    asm("v_mad_u64_u32 %0, vcc, %2, %4, %0\n\t"
        "v_mad_u64_u32 %1, vcc, %3, %4, %1"
        : "+v"(al[0]), "+v"(ah[0])
        : "v"(x0l), "v"(x0h), "n"(POSEIDON_MDS_DIAG0)
        : "vcc");
#pragma unroll
    for (int r = 0; r < W; r++) s[r] = gl::fold96(al[r], ah[r]);  // al + ah 2^32 mod p; al < 2^41 + X 2^32, ah < 2^41 + Y 2^32: X, Y leave the room
}

#endif  // POSEIDON_MDS_LAYER
#endif  // POSEIDON_MDS_NATURAL

// s-box layer + MDS layer; the round's own constants were added by the previous layer
__device__ __forceinline__ void full_round(uint64_t (&s)[W], const MdsOperands &ops, const uint32_t *__restrict__ xy) {
#pragma unroll
    for (int i = 0; i < W; i++) s[i] = gl::pow7(s[i]);
    mds_layer(s, ops, xy);
}

// All thirty rounds with their plain MDS layer — `poseidon_naive` (poseidon.rs:565-585), which the reference's own tests hold equal
// to Poseidon::poseidon. With the layer on the matrix cores a partial round (one s-box + one layer) is cheaper than its share of
// the "fast" partial rounds' 64-bit dot products: 2.61 G permutations/s against 2.52 with the blocked partial rounds between
// matrix-core full rounds and 2.25 on the vector ALU alone (tools/experiments/mds_mfma.hip, profiles/r03_poseidon_matrix_cores.jsonl).
__device__ __forceinline__ void permute(uint64_t (&s)[W], const MdsOperands &ops) {
    require_full_wave();
#pragma unroll
    for (int i = 0; i < W; i++) s[i] = gl::add_canonical(s[i], POSEIDON_ALL_ROUND_CONSTANTS[i]);
#pragma unroll 1
    for (int r = 0; r < HALF_FULL; r++) full_round(s, ops, POSEIDON_MDS_XY + 2 * W * r);
#pragma unroll 1
    for (int r = HALF_FULL; r < HALF_FULL + N_PARTIAL; r++) {
        s[0] = gl::pow7(s[0]);
        mds_layer(s, ops, POSEIDON_MDS_XY + 2 * W * r);
    }
#pragma unroll 1
    for (int r = HALF_FULL + N_PARTIAL; r < 2 * HALF_FULL + N_PARTIAL; r++) full_round(s, ops, POSEIDON_MDS_XY + 2 * W * r);
}

}  // namespace poseidon
