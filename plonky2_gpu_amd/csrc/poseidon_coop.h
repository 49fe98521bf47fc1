// poseidon_coop.h — ONE Poseidon permutation computed by a whole wavefront (latency version).
//
// The thread-per-permutation kernel (poseidon.h) is built for throughput: ~20k dependent VALU
// instructions per lane, i.e. ~60 us of latency. That is what a Fiat-Shamir transcript (serial by
// definition, iop/challenger.rs) and the top layers of a Merkle tree (a handful of nodes per
// layer) pay per permutation. Here the twelve state words live in twelve lanes:
//   * full rounds: constant + x^7 per lane, then the MDS matvec through LDS — every lane reads the
//     twelve words back (broadcast reads) and forms its own row with 24 v_mad_u64_u32;
//   * partial rounds: the "fast" recurrence (poseidon.rs:400-427)
//         u_r = sbox(s0) + rc_r;  d_r = c*u_r + sum_i s_i w_hat[r][i];  s_i += u_r v[r][i];  s0 = d_r
//     is unrolled in the u_q:   d_r = A_r + c*u_r + sum_{q<r} K[r][q] u_q,   K[r][q] = sum_i w_hat[r][i] v[q][i],
//     A_r = sum_i s_i(0) w_hat[r][i], and s_i(final) = s_i(0) + sum_q v[q][i] u_q. Lane r < 22 owns the lazy
//     accumulator of d_r, lane 32+i that of s_i; each round every lane adds ONE term
//     (table[q][lane] * u_q) to its accumulator, lane q's sum is complete and is broadcast. The serial
//     chain per round is one s-box, one accumulate, one reduction, one readlane.
// ~4k instructions per permutation instead of ~20k; same permutation bit for bit (tests compare
// with the reference's known answers).
#pragma once
#include "poseidon.h"

// gl::mul with its rare correction executed ALWAYS instead of behind a wave-uniform branch (three more instructions: with no borrow
// the mask is zero and they change nothing). The branch costs nothing when other waves fill the pipeline while it is resolved; the
// kernels here — the transcript's sponge, the top layers of a tree — are one wave alone on its SIMD, where every branch drains the
// vector pipeline (profiles/r05_mul_interleave.jsonl: 29 % of a multiplication chain at one wave per SIMD).
namespace gl {
__device__ __forceinline__ uint64_t mul_nb(uint64_t a, uint64_t b) {
    uint32_t al = (uint32_t)a, ah = (uint32_t)(a >> 32), bl = (uint32_t)b, bh = (uint32_t)(b >> 32);
    uint64_t r, c1;
    asm("v_mad_u64_u32 v[32:33], vcc, %2, %4, 0\n\t"
        "v_mad_u64_u32 v[34:35], vcc, %2, %5, 0\n\t"
        "v_mad_u64_u32 v[36:37], %1, %3, %4, v[34:35]\n\t"
        "v_mad_u64_u32 v[38:39], vcc, %3, %5, 0\n\t"
        "v_add_co_u32_e32 v33, vcc, v33, v36\n\t"
        "v_addc_co_u32_e32 v38, vcc, v38, v37, vcc\n\t"
        "v_addc_co_u32_e32 v39, vcc, 0, v39, vcc\n\t"
        "v_subb_co_u32_e64 v32, vcc, v32, v39, %1\n\t"
        "v_subbrev_co_u32_e32 v33, vcc, 0, v33, vcc\n\t"
        "v_cndmask_b32_e64 v42, 0, -1, vcc\n\t"
        "v_sub_co_u32_e32 v32, vcc, v32, v42\n\t"
        "v_subbrev_co_u32_e32 v33, vcc, 0, v33, vcc\n\t"
        "v_mad_u64_u32 v[32:33], vcc, v38, -1, v[32:33]\n\t"
        "v_cndmask_b32_e64 v42, 0, -1, vcc\n\t"
        "v_mad_u64_u32 %0, vcc, v42, 1, v[32:33]"
        : "=&v"(r), "=&s"(c1)
        : "v"(al), "v"(ah), "v"(bl), "v"(bh)
        : "vcc", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v42");
    return r;
}
__device__ __forceinline__ uint64_t pow7_nb(uint64_t x) {
    const uint64_t x2 = mul_nb(x, x), x4 = mul_nb(x2, x2), x3 = mul_nb(x, x2);
    return mul_nb(x3, x4);
}
}  // namespace gl

namespace poseidon_coop {

constexpr int LANES = 64;
constexpr int T0_ROWS = 11, T_ROWS = 22;

struct Tables {
    const uint64_t *t0;  // [11][64]: coefficient of the j-th word (j = 1..11) entering the partial rounds
    const uint64_t *t;   // [22][64]: coefficient of u_q
};

__device__ __forceinline__ uint64_t uniform64(uint64_t v) {
    uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
    uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    return (uint64_t)lo | ((uint64_t)hi << 32);
}

__device__ __forceinline__ uint64_t lane_value(uint64_t v, int lane) {
    uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, lane);
    uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), lane);
    return (uint64_t)lo | ((uint64_t)hi << 32);
}

// One-time table construction on the device (one wave; lane = table column).
__global__ void build_tables_kernel(uint64_t *t0, uint64_t *t) {
    const int lane = threadIdx.x;
    if (blockIdx.x || lane >= LANES) return;
    const uint64_t *M = POSEIDON_FAST_PARTIAL_ROUND_INITIAL_MATRIX, *WH = POSEIDON_FAST_PARTIAL_ROUND_W_HATS,
                   *VS = POSEIDON_FAST_PARTIAL_ROUND_VS;
    for (int j = 1; j <= 11; j++) {
        uint64_t v = 0;
        if (lane < T_ROWS) {
            for (int i = 1; i <= 11; i++) v = gl::add(v, gl::mul(M[(j - 1) * 11 + (i - 1)], WH[lane * 11 + (i - 1)]));
        } else if (lane > 32 && lane < 44) {
            v = M[(j - 1) * 11 + (lane - 32 - 1)];
        }
        t0[(j - 1) * LANES + lane] = gl::canon(v);
    }
    for (int q = 0; q < T_ROWS; q++) {
        uint64_t v = 0;
        if (lane < T_ROWS) {
            if (q < lane) {
                for (int i = 1; i <= 11; i++) v = gl::add(v, gl::mul(WH[lane * 11 + (i - 1)], VS[q * 11 + (i - 1)]));
            } else if (q == lane) {
                v = POSEIDON_MDS_CIRC[0] + POSEIDON_MDS_DIAG[0];
            }
        } else if (lane > 32 && lane < 44) {
            v = VS[q * 11 + (lane - 32 - 1)];
        }
        t[q * LANES + lane] = gl::canon(v);
    }
}

// lds: 12 u64 private to this wavefront. x = state word `lane` for lane < 12 (other lanes: anything).
// Every wavefront of the workgroup must call this together (it uses workgroup barriers).
__device__ __forceinline__ uint64_t permute(uint64_t x, const Tables &tb, uint64_t *lds) {
    const int lane = threadIdx.x & (LANES - 1);
    const bool active = lane < 12;
    const int l12 = active ? lane : 11;
    // this lane's row of circ(C) + diag(D) (poseidon.rs:174-194): out[r] = sum_j s[j] * C[(j - r) mod 12] + s[r] D[r]
    uint32_t coef[12];
#pragma unroll
    for (int j = 0; j < 12; j++)
        coef[j] = (uint32_t)POSEIDON_MDS_CIRC[(j - l12 + 12) % 12] + (j == l12 ? (uint32_t)POSEIDON_MDS_DIAG[l12] : 0u);

    auto full_round = [&](int round_ctr) {
        x = gl::add_canonical(x, POSEIDON_ALL_ROUND_CONSTANTS[l12 + 12 * round_ctr]);
        x = gl::pow7_nb(x);
        if (active) lds[lane] = x;
        __syncthreads();
        uint64_t al = 0, ah = 0;
#pragma unroll
        for (int j = 0; j < 12; j++) {
            uint64_t w = lds[j];
            al += (w & 0xFFFFFFFFull) * coef[j];
            ah += (w >> 32) * coef[j];
        }
        __syncthreads();
        uint64_t l = al + (ah << 32);
        uint32_t h = (uint32_t)(ah >> 32) + (l < al ? 1u : 0u);
        x = gl::reduce96(l, h);
    };

#pragma unroll 1
    for (int r = 0; r < poseidon::HALF_FULL; r++) full_round(r);

    // partial_first_constant_layer (poseidon.rs:312-320), then mds_partial_layer_init folded into the tables
    x = gl::add_canonical(x, POSEIDON_FAST_PARTIAL_FIRST_ROUND_CONSTANT[l12]);
    if (active) lds[lane] = x;
    __syncthreads();
    const uint64_t s0 = lds[0];
    gl::DotAcc acc;
#pragma unroll 1
    for (int j = 1; j <= 11; j++) {
        uint64_t sj = uniform64(lds[j]);
        asm volatile("s_nop 2" : "+s"(sj));  // v_readfirstlane -> SGPR read inside inline asm (see fri.hip)
        gl::dot_term(acc, tb.t0[(j - 1) * LANES + lane], sj);
    }
    __syncthreads();
    uint64_t u = gl::add_canonical(gl::pow7_nb(s0), POSEIDON_FAST_PARTIAL_ROUND_CONSTANTS[0]);
    uint64_t d = 0, dq = 0;
#pragma unroll 1
    for (int q = 0; q < T_ROWS; q++) {
        uint64_t us = uniform64(u);
        asm volatile("s_nop 2" : "+s"(us));
        gl::dot_term(acc, tb.t[q * LANES + lane], us);
        d = gl::dot_finish(acc);
        dq = lane_value(d, q);
        if (q + 1 < T_ROWS) u = gl::add_canonical(gl::pow7_nb(dq), POSEIDON_FAST_PARTIAL_ROUND_CONSTANTS[q + 1]);
    }
    // gather: word 0 = d_21, word i = the accumulator of lane 32+i
    if (lane > 32 && lane < 44) lds[lane - 32] = d;
    if (lane == 0) lds[0] = dq;
    __syncthreads();
    x = lds[l12];
    __syncthreads();

#pragma unroll 1
    for (int r = 0; r < poseidon::HALF_FULL; r++) full_round(poseidon::HALF_FULL + poseidon::N_PARTIAL + r);
    return x;
}

}  // namespace poseidon_coop
