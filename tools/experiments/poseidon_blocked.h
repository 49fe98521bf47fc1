// poseidon_blocked.h — EXPERIMENT (round 4), not part of the product: the twenty-two partial rounds of the matrix-core Poseidon with
// twenty of them as five BLOCKS of four (one pass over the matrix cores per block), on top of csrc/poseidon.h.
//
// Outcome (profiles/r04_poseidon_blocked_partial_rounds.jsonl, LABNOTES section 11): bit-exact on 2^22 states x 32 permutations and in
// the CPU model (tests/test_poseidon_matrix_model.py); 12.4 k instead of 14.8 k vector instructions per permutation in the code
// (-16 %) — and NOT faster: a block needs about 190 registers (fifteen rows x two 64-bit sums + eight byte planes + four digit planes
// + the chains in flight), i.e. two waves per SIMD, where the hand-scheduled field arithmetic of the full rounds (asm blocks the
// compiler cannot interleave) runs 25 % slower than at four waves. 6.03 ms per 2^22 x 4 permutations against 5.92; running the chains
// twice to fit three waves (168 registers): 6.01 against 6.14 on the same device. Kept so that the numbers can be reproduced:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form -I plonky2_gpu_amd/csrc -I tools/experiments \
//         [-DBLOCK_WAVES=3] tools/experiments/mds_mfma.hip -o mds_mfma
#pragma once
#include <type_traits>
#include "poseidon.h"
#include "poseidon_block_constants.h"

namespace poseidon {

typedef uint32_t v16u32 __attribute__((ext_vector_type(16)));

// ---- T = 4 partial rounds as ONE pass over the matrix cores (tables and derivation: tools/gen_poseidon_block_tables.py) ---------
// With N = M Z (the MDS matrix without its column 0: element 0 goes through the s-box) and m0 = M e_0, four partial rounds are
//     v_4 = N^4 v_0 + sum_t y_t N^(3-t) m0 + constants,     x_t = (N^t v_0)[0] + sum_{u<t} y_u (N^(t-1-u) m0)[0] + constant,  y_t = x_t^7.
// The fifteen linear forms of v_0 (twelve rows of N^4, row 0 of N, N^2, N^3) are fifteen of the sixteen result rows a lane gets from
// one matrix instruction. Their entries have up to 29 bits: four SIGNED base-256 digit planes A_p; products of equal weight
// 256^(p + k) (digit plane p, state byte plane k) are chained in the accumulator, eleven chains D_0 .. D_10 from zero (signed sums,
// below 2^20 in magnitude), 32 matrix instructions — as many as four plain layers — but the vector ALU unpacks
// the state, recombines the planes and reduces ONCE per block:
//     E_j = D_2j + (D_2j+1 << 8)                    (32-bit)
//     G0 = E0 + E1 2^16 + Hl 2^32,  G1 = E2 + E3 2^16 + Hh 2^32,  G2 = E4 + E5 2^16        (64-bit; value = G0 + G1 2^32 + G2 2^64)
//     al = G0 - G2,  ah = G1 + G2                     (2^64 = 2^32 - 1 mod p), then the rank-one terms y_t x weight as two multiply-adds,
//     one gl::fold96 per row.
// (Hl, Hh) and the row's entry e in the matrix's column 0 — zero in every power of N; the kernel feeds the signed byte 1 there in plane
// 0 — carry the additive constants and the offset of the signed bytes (Hl 2^32 + Hh 2^64 + e = both mod p), chosen by the generator so
// that every sum provably stays inside 64 bits; tests/test_poseidon_matrix_model.py executes this very procedure on the tables.
struct BlockA {
    v4i32 p[POSEIDON_BLOCK_T];
};

// this lane's rows of the block's A operand (a per-lane table look-up: call with every lane active)
__device__ __forceinline__ BlockA block_a(int block) {
    const uint32_t lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const uint32_t m = lane & 31, h = lane >> 5;
    const uint32_t q = (m & 3) + 4 * (m >> 3);  // as mds_operands: result register q of either half is logical row q
    const bool on = ((m >> 2) & 1) == h;
    const v4i32 *tab = reinterpret_cast<const v4i32 *>(POSEIDON_BLOCK_A) + ((uint32_t)block * 16 + q) * POSEIDON_BLOCK_T;
    BlockA a;
#pragma unroll
    for (int p = 0; p < POSEIDON_BLOCK_T; p++) {
        const v4i32 w = tab[p];
#pragma unroll
        for (int c = 0; c < 4; c++) a.p[p][c] = on ? w[c] : 0;
    }
    return a;
}

// acc += a * c as one v_mad_u64_u32 (plain C: the compiler selects the instruction itself and, unlike an asm statement, needs no
// wait state behind it)
__device__ __forceinline__ void mad32(uint64_t &acc, uint32_t a, uint32_t c) { acc += (uint64_t)a * c; }

__device__ __forceinline__ void partial_block(uint64_t (&s)[W], int block) {
    static_assert(POSEIDON_BLOCK_T == 4 && POSEIDON_BLOCK_PLANES == 11 && POSEIDON_BLOCK_ROWS == 15 && POSEIDON_BLOCK_BIAS == 0, "written for T = 4, signed planes");
    BlockA A = block_a(block);
    // the block's thirty high-dword constants as two 64-byte scalar loads, issued before anything needs them
    const v16u32 HL = *reinterpret_cast<const v16u32 *>(POSEIDON_BLOCK_H + (uint32_t)block * 32);
    const v16u32 HH = *reinterpret_cast<const v16u32 *>(POSEIDON_BLOCK_H + (uint32_t)block * 32 + 16);
    // T[k] = byte k of words 0-3 | 4-7 | 8-11 as byte - 128 | (never written: meets zero columns of A). Element 0 enters the block only
    // through the s-box (column 0 of every power of N is zero); in its place goes the constant word 0x8080808080808081 — the signed byte
    // 1 in plane 0, zeros above — so that column 0 of the A operand is a free additive constant per row (the generator's e).
    const uint64_t x0 = s[0];
    v4i32 T[8];
#pragma unroll
    for (int G = 0; G < 3; G++)
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const uint32_t r0 = G == 0 ? (h == 0 ? 0x80808081u : 0x80808080u) : (uint32_t)(s[4 * G] >> (32 * h));
            const uint32_t r1 = (uint32_t)(s[4 * G + 1] >> (32 * h));
            const uint32_t r2 = (uint32_t)(s[4 * G + 2] >> (32 * h)), r3 = (uint32_t)(s[4 * G + 3] >> (32 * h));
            const uint32_t a01 = __builtin_amdgcn_perm(r1, r0, 0x05010400u), c01 = __builtin_amdgcn_perm(r1, r0, 0x07030602u);
            const uint32_t a23 = __builtin_amdgcn_perm(r3, r2, 0x05010400u), c23 = __builtin_amdgcn_perm(r3, r2, 0x07030602u);
            T[4 * h + 0][G] = (int)(__builtin_amdgcn_perm(a23, a01, 0x05040100u) ^ 0x80808080u);
            T[4 * h + 1][G] = (int)(__builtin_amdgcn_perm(a23, a01, 0x07060302u) ^ 0x80808080u);
            T[4 * h + 2][G] = (int)(__builtin_amdgcn_perm(c23, c01, 0x05040100u) ^ 0x80808080u);
            T[4 * h + 3][G] = (int)(__builtin_amdgcn_perm(c23, c01, 0x07060302u) ^ 0x80808080u);
        }
    // chain w = sum over digit plane p and byte plane k = w - p, from zero: SIGNED plane sums, |chain| < 2^20
    auto chain = [&](int w) {
        v16i32 d = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int p = 0; p < POSEIDON_BLOCK_T; p++) {
            const int k = w - p;
            if (k >= 0 && k < 8) d = __builtin_amdgcn_mfma_i32_32x32x32_i8(A.p[p], T[k], d, 0, 0, 0);
        }
        return d;
    };
    int32_t k16 = 65536;  // opaque: e * k16 stays one 64-bit multiply-add instead of a 64-bit shift and a carry add
    asm volatile("" : "+s"(k16));
    // group sum = e_lo + e_hi 2^16 + H 2^32 in wrapping 64-bit arithmetic (the generator guarantees that al and ah, the sums that are
    // used, are true non-negative 64-bit numbers)
    auto group = [&](int32_t lo, int32_t hi, uint32_t H) { return (uint64_t)((int64_t)lo + (int64_t)hi * k16) + ((uint64_t)H << 32); };
    // The chains are run TWICE (the matrix pipe is idle most of the time; the vector ALU is what is short): a pass keeps al, ah of the
    // rows in [Q0, Q1) only, so that the block fits three waves per SIMD. Pass 1: the x rows (12-14) and rows 0-4, then the s-box chain
    // and rows 0-4 finished; pass 2: rows 5-11.
    // Order and fences: top group first (its sum enters both al and ah), then the bottom group, then the middle one in place of the
    // top group's sum; the scheduler must not start a group's chains before the previous group's results are packed.
    auto rows = [&](auto first_c, auto count_c, auto extra_c, uint64_t *al, uint64_t *ah) {
        // the rows [first, first + count) and, with extra, the three x rows 12-14 behind them
        constexpr int Q0 = decltype(first_c)::value, NQ = decltype(count_c)::value, N = NQ + (decltype(extra_c)::value ? 3 : 0);
        auto row = [](int i) { return i < NQ ? Q0 + i : 12 + (i - NQ); };
        // a pass's chains are its own: without this the compiler computes the eleven chains once and keeps all 176 registers alive
        asm volatile("" : "+v"(A.p[0]), "+v"(A.p[1]), "+v"(A.p[2]), "+v"(A.p[3]));
        auto pair = [&](int j, int32_t *e) {
            const v16i32 lo = chain(2 * j), hi = chain(2 * j + 1);
#pragma unroll
            for (int i = 0; i < N; i++) e[i] = (int32_t)((uint32_t)lo[row(i)] + ((uint32_t)hi[row(i)] << 8));
        };
        {
            int32_t e4[N];
            pair(4, e4);
            const v16i32 D10 = chain(10);
#pragma unroll
            for (int i = 0; i < N; i++) ah[i] = group(e4[i], D10[row(i)], 0);  // G2
        }
        __builtin_amdgcn_sched_barrier(0);
        {
            int32_t e0[N], e1[N];
            pair(0, e0);
            __builtin_amdgcn_sched_barrier(0);
            pair(1, e1);
#pragma unroll
            for (int i = 0; i < N; i++) al[i] = group(e0[i], e1[i], HL[row(i)]) - ah[i];
        }
        __builtin_amdgcn_sched_barrier(0);
        {
            int32_t e2[N], e3[N];
            pair(2, e2);
            __builtin_amdgcn_sched_barrier(0);
            pair(3, e3);
#pragma unroll
            for (int i = 0; i < N; i++) ah[i] += group(e2[i], e3[i], HH[row(i)]);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    // the weights of the rank-one terms: four 64-byte scalar loads
    const uint32_t *__restrict__ UX = POSEIDON_BLOCK_U + (uint32_t)block * 64;
    const v16u32 U0 = *reinterpret_cast<const v16u32 *>(UX), U1 = *reinterpret_cast<const v16u32 *>(UX + 16);
    const v16u32 U2 = *reinterpret_cast<const v16u32 *>(UX + 32), XC = *reinterpret_cast<const v16u32 *>(UX + 48);
    auto weight = [&](int j, int q) { const int i = j * W + q; return i < 16 ? U0[i] : i < 32 ? U1[i - 16] : U2[i - 32]; };
    uint32_t yl[POSEIDON_BLOCK_T], yh[POSEIDON_BLOCK_T];
    auto finish = [&](int q, uint64_t a_l, uint64_t a_h) {
#pragma unroll
        for (int t = 0; t < POSEIDON_BLOCK_T; t++) {
            mad32(a_l, yl[t], weight(POSEIDON_BLOCK_T - 1 - t, q));
            mad32(a_h, yh[t], weight(POSEIDON_BLOCK_T - 1 - t, q));
        }
        s[q] = gl::fold96(a_l, a_h);
    };
    constexpr int SPLIT = 5;
    {
        uint64_t al[SPLIT + 3], ah[SPLIT + 3];
        rows(std::integral_constant<int, 0>{}, std::integral_constant<int, SPLIT>{}, std::true_type{}, al, ah);
        // the s-box chain: x_0 = element 0 as it came in, x_t from row 11 + t and the earlier y's
        uint64_t x = x0;
#pragma unroll
        for (int t = 0; t < POSEIDON_BLOCK_T; t++) {
            if (t) {
#pragma unroll
                for (int u = 0; u < t; u++) {
                    mad32(al[SPLIT + t - 1], yl[u], XC[t * POSEIDON_BLOCK_T + u]);
                    mad32(ah[SPLIT + t - 1], yh[u], XC[t * POSEIDON_BLOCK_T + u]);
                }
                x = gl::fold96(al[SPLIT + t - 1], ah[SPLIT + t - 1]);
            }
            const uint64_t y = gl::pow7(x);
            yl[t] = (uint32_t)y, yh[t] = (uint32_t)(y >> 32);
        }
#pragma unroll
        for (int q = 0; q < SPLIT; q++) finish(q, al[q], ah[q]);
    }
    {
        uint64_t al[W - SPLIT], ah[W - SPLIT];
        rows(std::integral_constant<int, SPLIT>{}, std::integral_constant<int, W - SPLIT>{}, std::false_type{}, al, ah);
#pragma unroll
        for (int q = SPLIT; q < W; q++) finish(q, al[q - SPLIT], ah[q - SPLIT]);
    }
}

// full rounds and the first two partial rounds as plain layers, the other twenty partial rounds as five blocks of four
__device__ __forceinline__ void permute_blocked(uint64_t (&s)[W], const MdsOperands &ops) {
    require_full_wave();
#pragma unroll
    for (int i = 0; i < W; i++) s[i] = gl::add_canonical(s[i], POSEIDON_ALL_ROUND_CONSTANTS[i]);
#pragma unroll 1
    for (int r = 0; r < HALF_FULL; r++) full_round(s, ops, POSEIDON_MDS_XY + 2 * W * r);
#pragma unroll 1
    for (int r = HALF_FULL; r < POSEIDON_BLOCK_FIRST_ROUND; r++) {
        s[0] = gl::pow7(s[0]);
        mds_layer(s, ops, POSEIDON_MDS_XY + 2 * W * r);
    }
    static_assert(POSEIDON_BLOCK_FIRST_ROUND + POSEIDON_BLOCK_T * POSEIDON_BLOCK_COUNT == HALF_FULL + N_PARTIAL, "the blocks end where the last full rounds begin");
#pragma unroll 1
    for (int b = 0; b < POSEIDON_BLOCK_COUNT; b++) partial_block(s, b);
#pragma unroll 1
    for (int r = HALF_FULL + N_PARTIAL; r < 2 * HALF_FULL + N_PARTIAL; r++) full_round(s, ops, POSEIDON_MDS_XY + 2 * W * r);
}

}  // namespace poseidon
