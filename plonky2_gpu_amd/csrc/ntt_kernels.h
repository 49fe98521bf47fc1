// ntt_kernels.h — what the NTT pass kernels share: the pass description the planner fills in, compile-time loops,
// twiddle look-ups, the in-register radix butterflies and the two kinds of synchronisation. Included by ntt.hip (planner,
// tile kernels) and ntt_direct.hip (direct passes); everything is inline device code in an internal namespace.
#pragma once
#include <atomic>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "gl_field.h"

namespace plonky2_hip {
namespace nttk {

enum : uint32_t { F_LOAD_ROWS = 1, F_STORE_ROWS = 2, F_NATURAL = 4, F_INVERSE = 8, F_COSET = 16,
                  F_WIDE = 32,
                  F_RAW_OUT = 64,
                  F_FINAL_COL = 128 };  // direct column pass that is the LAST pass of a transform: no inter-pass twiddle, canonical output  // the pass feeds another pass: its output need not be canonical (any u64 representative is a legal input)  // F_WIDE: the planner laid the pass out for tiles of 2^(LOGE+1) elements (ntt_pass_wave_kernel, LOGW = 4)

struct PassParams {
    const uint64_t *src;
    uint64_t *dst;
    const uint64_t *twl;  // w_{2^24}^e, e < 4096
    const uint64_t *twh;  // w_{2^12}^e, e < 4096
    const uint64_t *cs_hi;  // coset scale tables (F_COSET): s_r^(e<<10), per coset r
    const uint64_t *cs_lo;  // s_r^e, e < 1024
    uint64_t in_sa, in_sb, in_sz, in_t, in_m;
    uint64_t out_sa, out_sb, out_sz, out_t, out_m;
    uint64_t out_c;      // direct column pass only: stride between the columns of a tile in the output (0 = 1 = adjacent; N1 with out_m = 1
                         // writes the tile TRANSPOSED — the first pass of the two-pass plan for 2^22, ntt.hip)
    uint64_t scale;      // multiplied into every output (1 = none)
    uint64_t chain_scale;  // folded into the inter-pass twiddle chain start (1 = none): n^-1 of the inverse
    uint32_t logt;       // log2 T
    uint32_t t_limit;    // valid range of b*T + t
    uint32_t flags;
    uint32_t log_n;      // polynomial length (index flip for the inverse)
    uint32_t tw_hi;      // inter-pass twiddle root = w_{2^tw_hi}
    uint32_t cs_hi_len;  // entries per coset in cs_hi
    uint32_t rate_bits;  // F_COSET: blockIdx.z = coset r, written to block bitrev(r)
    uint32_t row_shift;  // inverse natural-order row pass: rotate the row tile by one so that the
                         // flipped 64-byte output segments are aligned (t_limit is a power of two)
};

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

constexpr int brev_c(int x, int bits) {
    int r = 0;
    for (int i = 0; i < bits; i++) r |= ((x >> i) & 1) << (bits - 1 - i);
    return r;
}

__device__ __forceinline__ uint32_t brev_rt(uint32_t x, int bits) { return bits ? (__brev(x) >> (32 - bits)) : 0; }

__device__ __forceinline__ uint32_t phys(uint32_t idx, uint32_t logt) { return idx + ((idx >> (4 + logt)) << logt); }

// w_{2^hi}^x through the two-level table of w_{2^24}
__device__ __forceinline__ uint64_t wpow(const PassParams &p, uint64_t x) {
    uint32_t e = (uint32_t)(x << (24 - p.tw_hi)) & 0xFFFFFFu;
    uint64_t h = p.twh[e >> 12];
    uint32_t lo = e & 4095u;
    return lo ? gl::mul(h, p.twl[lo]) : h;
}

// Inter-pass twiddle of the thread that holds, after the pass's last radix-16 round, the outputs
// k1 = bitrev4(i)*(R/16) + kr of column L: w^(L*k1) = c * step^bitrev4(i) with c = w^(L*kr) (times the coset power
// s_r^L and the inverse's 1/n where they apply) and step = w^(L*R/16). Four table look-ups in global memory.
template <int LOGT, int LOGR>
__device__ __forceinline__ void twiddle_chain(const PassParams &p, uint32_t tid, uint32_t b, uint32_t z, uint64_t &c, uint64_t &step) {
    constexpr uint32_t TMASK = (1u << LOGT) - 1;
    uint32_t l = tid & TMASK, rest = tid >> LOGT;
    uint64_t L = (uint64_t)b * (1u << LOGT) + l;
    uint32_t kr = brev_rt(rest, LOGR - 4);
    c = wpow(p, L * kr);
    if (p.flags & F_COSET) {
        // fold s_r^L (coset shift power of the low index) into the chain start
        uint64_t sl = gl::mul(p.cs_hi[z * p.cs_hi_len + (uint32_t)(L >> 10)], p.cs_lo[z * 1024 + (uint32_t)(L & 1023)]);
        c = gl::mul(c, sl);
    }
    if (p.chain_scale != 1) c = gl::mul(c, p.chain_scale);
    step = wpow(p, L << (LOGR - 4));
}

// In-register radix-2^D DIF butterfly on v[BASE .. BASE+2^D): output slot i holds frequency
// bitrev_D(i). Stage twiddles w_{2^(s+1)}^j = 2^(39*j*(32>>s)) are multiply-free. radix_dif_stage is one of its D stages
// (s = D-1 first), for callers that put other work between the stages.
#ifndef RARE_GB
#define RARE_GB 4
#endif
#ifndef RARE_FENCE_MASK
#define RARE_FENCE_MASK 0   // what may be scheduled across the fence in front of a group's mask ORs (0: nothing)
#endif
#ifndef RARE_STAGE_COMBINED
#define RARE_STAGE_COMBINED 1
#endif
// NBLK adjacent blocks of 2^D slots starting at BASE go through stage s TOGETHER (their butterflies are independent: larger groups)
template <int D, int BASE, int s, int NBLK = 1>
__device__ __forceinline__ void radix_dif_stage(uint64_t (&v)[16]) {
    constexpr int half = 1 << s;
    constexpr int NBP = (1 << D) / 2;   // butterflies per block
    constexpr int NB = NBP * NBLK;
    // The rare paths of the field operations are DEFERRED (gl_field.h, bfly_f / mul_pow2_f): the sums and differences of up to
    // four butterflies run their fast paths back to back, the eight masks are OR-ed and ONE branch guards the corrections; then the
    // shift twiddles of the stage the same way. A stage of a radix-16 butterfly has three or four branches instead of twenty-four.
#if RARE_STAGE_COMBINED
    // ONE group per stage: the butterflies and then the shift twiddles of their differences, all fast paths back to back. A
    // difference whose second borrow is pending (true d = d_fast - e) goes through its shift as it is: the shift is linear and exact
    // for any representative, so the true result is shift(d_fast) - e 2^K, a compile-time constant to subtract in the flagged lanes.
    if constexpr (s > 0 && NB <= 8) {
        gl::rare_mask fa[NB], fs[NB], fm[NB];
        static_for<0, NB>([&](auto B_) {
            constexpr int b = decltype(B_)::value, bb = b % NBP;
            constexpr int i0 = BASE + (b / NBP) * (1 << D) + (bb / half) * 2 * half + (bb % half), i1 = i0 + half;
            constexpr int K = (39 * (bb % half) * (32 >> s)) % 192;
            gl::bfly_f<(K >= 96)>(v[i0], v[i1], v[i0], v[i1], fa[b], fs[b]);
        });
        static_for<0, NB>([&](auto B_) {
            constexpr int b = decltype(B_)::value, bb = b % NBP;
            constexpr int i1 = BASE + (b / NBP) * (1 << D) + (bb / half) * 2 * half + (bb % half) + half;
            constexpr int K = (39 * (bb % half) * (32 >> s)) % 192, KK = K >= 96 ? K - 96 : K;
            v[i1] = gl::mul_pow2_f<KK>(v[i1], fm[b]);
        });
        __builtin_amdgcn_sched_barrier(RARE_FENCE_MASK);
        gl::rare_mask any = 0;
        static_for<0, NB>([&](auto B_) { constexpr int b = decltype(B_)::value; any |= fa[b] | fs[b] | fm[b]; });
        if (GL_RARE_ANY(any)) {
            static_for<0, NB>([&](auto B_) {
                constexpr int b = decltype(B_)::value, bb = b % NBP;
                constexpr int i0 = BASE + (b / NBP) * (1 << D) + (bb / half) * 2 * half + (bb % half), i1 = i0 + half;
                constexpr int K = (39 * (bb % half) * (32 >> s)) % 192, KK = K >= 96 ? K - 96 : K;
                v[i0] = gl::add_fix(v[i0], fa[b]);
                v[i1] = gl::mul_pow2_fix<KK>(v[i1], fm[b]);
                #ifdef RARE_STAGE_WRONG_CONSTANT   // negative control of the tests only: the build must FAIL tests/test_gpu_ntt.py's rare-path test
                v[i1] = gl::sub(v[i1], gl::masked_const<gl::eps_times_pow2(KK) + 1>(fs[b]));
#else
                v[i1] = gl::sub(v[i1], gl::masked_const<gl::eps_times_pow2(KK)>(fs[b]));   // KK = 0: e itself
#endif
            });
        }
        return;
    }
#endif
    constexpr int GB = NB >= RARE_GB ? RARE_GB : NB;  // butterflies per group
    static_assert(NB % GB == 0, "whole groups");
    static_for<0, NB / GB>([&](auto G_) {
        constexpr int g0 = decltype(G_)::value * GB;
        gl::rare_mask fa[GB], fs[GB];
        static_for<0, GB>([&](auto B_) {
            constexpr int b = g0 + decltype(B_)::value, k = decltype(B_)::value, bb = b % NBP;
            constexpr int i0 = BASE + (b / NBP) * (1 << D) + (bb / half) * 2 * half + (bb % half), i1 = i0 + half;
            constexpr int K = (39 * (bb % half) * (32 >> s)) % 192;
            // 2^96 = -1: a twiddle 2^K with K >= 96 is -(2^(K-96)); the sign is absorbed by swapping the operands of the subtraction
            gl::bfly_f<(K >= 96)>(v[i0], v[i1], v[i0], v[i1], fa[k], fs[k]);
        });
        __builtin_amdgcn_sched_barrier(RARE_FENCE_MASK);  // the ORs behind the last operation, see rare_group
        gl::rare_mask any = 0;
        static_for<0, GB>([&](auto B_) { any |= fa[decltype(B_)::value] | fs[decltype(B_)::value]; });
        if (GL_RARE_ANY(any)) {
            static_for<0, GB>([&](auto B_) {
                constexpr int b = g0 + decltype(B_)::value, k = decltype(B_)::value, bb = b % NBP;
                constexpr int i0 = BASE + (b / NBP) * (1 << D) + (bb / half) * 2 * half + (bb % half), i1 = i0 + half;
                v[i0] = gl::add_fix(v[i0], fa[k]);
                v[i1] = gl::sub_fix(v[i1], fs[k]);
            });
        }
    });
    if constexpr (s > 0) {  // stage 0 has no twiddles (K = 0 for every butterfly)
        static_assert(NB <= 8, "the shift twiddles of a stage form one group");
        gl::rare_mask fm[NB];
        static_for<0, NB>([&](auto B_) {
            constexpr int b = decltype(B_)::value, bb = b % NBP;
            constexpr int i1 = BASE + (b / NBP) * (1 << D) + (bb / half) * 2 * half + (bb % half) + half;
            constexpr int K = (39 * (bb % half) * (32 >> s)) % 192, KK = K >= 96 ? K - 96 : K;
            v[i1] = gl::mul_pow2_f<KK>(v[i1], fm[b]);
        });
        __builtin_amdgcn_sched_barrier(RARE_FENCE_MASK);
        gl::rare_mask any = 0;
        static_for<0, NB>([&](auto B_) { any |= fm[decltype(B_)::value]; });
        if (GL_RARE_ANY(any)) {
            static_for<0, NB>([&](auto B_) {
                constexpr int b = decltype(B_)::value, bb = b % NBP;
                constexpr int i1 = BASE + (b / NBP) * (1 << D) + (bb / half) * 2 * half + (bb % half) + half;
                constexpr int K = (39 * (bb % half) * (32 >> s)) % 192, KK = K >= 96 ? K - 96 : K;
                v[i1] = gl::mul_pow2_fix<KK>(v[i1], fm[b]);
            });
        }
    }
}

// N independent field operations with their rare paths deferred (gl_field.h): one(I, mask&) runs the fast path of operation I,
// fix(I, mask) corrects its result; the corrections sit behind ONE branch. N <= 8 keeps the masks within sixteen scalar registers.
template <int N, class One, class Fix>
__device__ __forceinline__ void rare_group(One &&one, Fix &&fix) {
    static_assert(N >= 1 && N <= 8, "a group's masks must fit the scalar registers a kernel has to spare");
    gl::rare_mask f[N];
    static_for<0, N>([&](auto I_) { one(I_, f[decltype(I_)::value]); });
    // the scalar ORs wait for the vector instruction that wrote their mask: left to itself the scheduler puts each OR right behind
    // its producer (N stalls); behind this fence they all come after the last operation (one)
    __builtin_amdgcn_sched_barrier(RARE_FENCE_MASK);
    gl::rare_mask any = 0;
    static_for<0, N>([&](auto I_) { any |= f[decltype(I_)::value]; });
    if (GL_RARE_ANY(any)) static_for<0, N>([&](auto I_) { fix(I_, f[decltype(I_)::value]); });
}

// v[idx(k)] *= w(k) for k = LO .. HI-1, eight multiplications per group; after(k) runs behind multiplication k (the passes put the
// tail steps of the previous tile there)
template <int LO, int HI, class Idx, class W, class After>
__device__ __forceinline__ void mul_run(uint64_t (&v)[16], Idx &&idx, W &&w, After &&after) {
    constexpr int N = HI - LO;
    static_for<0, (N + 7) / 8>([&](auto G_) {
        constexpr int k0 = LO + 8 * decltype(G_)::value, n = (HI - k0) < 8 ? (HI - k0) : 8;
        rare_group<n>(
            [&](auto I_, gl::rare_mask &f) {
                constexpr int k = k0 + decltype(I_)::value;
                auto K_ = std::integral_constant<int, k>{};
                v[idx(K_)] = gl::mul_f(v[idx(K_)], w(K_), f);
                after(K_);
            },
            [&](auto I_, gl::rare_mask f) {
                constexpr int k = k0 + decltype(I_)::value;
                auto K_ = std::integral_constant<int, k>{};
                v[idx(K_)] = gl::mul_fix(v[idx(K_)], f);
            });
    });
}

template <int D, int BASE>
__device__ __forceinline__ void radix_dif(uint64_t (&v)[16]) {
    static_for<0, D>([&](auto S_) { radix_dif_stage<D, BASE, D - 1 - decltype(S_)::value>(v); });
}
// radix 2^D on all 16 >> D blocks of the sixteen registers at once, stage by stage
template <int D>
__device__ __forceinline__ void radix_dif_blocks(uint64_t (&v)[16]) {
    static_for<0, D>([&](auto S_) { radix_dif_stage<D, 0, D - 1 - decltype(S_)::value, (16 >> D)>(v); });
}

// The last radix 4 of a 1024-point row WITH the twiddles in front of it as shifts (row pass with natural-order output). Sixteen
// registers v[4 kblo + q]: q = input index of the radix 4, kB = 4 KBHI + kblo = output index of the radix 16 before it; the twiddle
// between them is w_64^(q kB), and every 64th root of unity of this field is a power of two (w_64 = 2^39, 2^96 = -1): a shift by
// (39 q kB) mod 96 and a sign instead of a table look-up and a twelve-instruction multiplication. The signs cost nothing: a negated
// subtrahend swaps the roles of a butterfly's two results, and the sign of x1 (the minuend of its pair) is handed on to the second
// stage, where x1's sum and difference are the subtrahends. KBHI must be a compile-time constant: the caller switches on it
// (it is wave-uniform there). Rare paths deferred, three groups; a pending correction of the value that goes through the stage's
// own shift (2^48) is applied behind it as the constant e 2^48 (see radix_dif_stage).
template <int KBHI>
__device__ __forceinline__ void shift_twiddles_radix4(uint64_t (&v)[16]) {
    constexpr auto KQ = [](int kblo, int q) { return (39 * q * (4 * KBHI + kblo)) % 192; };
    constexpr auto NEGQ = [](int kblo, int q) { return (39 * q * (4 * KBHI + kblo)) % 192 >= 96; };
    {   // the twiddles of q = 1, 2, 3 (their signs are NEGQ)
        gl::rare_mask fm[12];
        static_for<0, 12>([&](auto I_) {
            constexpr int i = decltype(I_)::value, kblo = i / 3, q = 1 + i % 3, KK = KQ(kblo, q) % 96;
            v[4 * kblo + q] = gl::mul_pow2_f<KK>(v[4 * kblo + q], fm[i]);
        });
        __builtin_amdgcn_sched_barrier(RARE_FENCE_MASK);
        gl::rare_mask any = 0;
        static_for<0, 12>([&](auto I_) { any |= fm[decltype(I_)::value]; });
        if (GL_RARE_ANY(any))
            static_for<0, 12>([&](auto I_) {
                constexpr int i = decltype(I_)::value, kblo = i / 3, q = 1 + i % 3, KK = KQ(kblo, q) % 96;
                v[4 * kblo + q] = gl::mul_pow2_fix<KK>(v[4 * kblo + q], fm[i]);
            });
    }
    // butterfly (a, c) -> (a + C, a - C) in slots (i0, i1), C = -c when CNEG: then slot i0 takes the difference and i1 the sum.
    // f0 / f1: the masks of the values written to i0 / i1
    auto bfly = [&](auto CNEG_, auto I0_, auto I1_, gl::rare_mask &f0, gl::rare_mask &f1) {
        constexpr bool cneg = decltype(CNEG_)::value;
        constexpr int i0 = decltype(I0_)::value, i1 = decltype(I1_)::value;
        if constexpr (cneg) gl::bfly_f<false>(v[i0], v[i1], v[i1], v[i0], f1, f0);
        else gl::bfly_f<false>(v[i0], v[i1], v[i0], v[i1], f0, f1);
    };
    // correction of slot I: it holds a sum (IS_SUM) or a difference, and went through a shift by SHIFT afterwards (0: none)
    auto fix = [&](auto IS_SUM_, auto SHIFT_, auto I_, gl::rare_mask f, gl::rare_mask fshift) {
        constexpr bool is_sum = decltype(IS_SUM_)::value;
        constexpr int shift = decltype(SHIFT_)::value, i = decltype(I_)::value;
        if constexpr (shift == 0) {
            v[i] = is_sum ? gl::add_fix(v[i], f) : gl::sub_fix(v[i], f);
        } else {
            v[i] = gl::mul_pow2_fix<shift>(v[i], fshift);
            const uint64_t c = gl::masked_const<gl::eps_times_pow2(shift)>(f);
            v[i] = is_sum ? gl::add(v[i], c) : gl::sub(v[i], c);
        }
    };
    using std::integral_constant;
    using std::bool_constant;
    {   // first stage: pairs (0, 2) and (1, 3) of every block, then 2^48 on slot 3
        gl::rare_mask f0[8], f1[8], fm[4];
        static_for<0, 4>([&](auto B_) {
            constexpr int kblo = decltype(B_)::value, b = 4 * kblo;
            bfly(bool_constant<NEGQ(kblo, 2)>{}, integral_constant<int, b>{}, integral_constant<int, b + 2>{}, f0[2 * kblo], f1[2 * kblo]);
            bfly(bool_constant<NEGQ(kblo, 1) != NEGQ(kblo, 3)>{}, integral_constant<int, b + 1>{}, integral_constant<int, b + 3>{}, f0[2 * kblo + 1], f1[2 * kblo + 1]);
        });
        static_for<0, 4>([&](auto B_) {
            constexpr int kblo = decltype(B_)::value;
            v[4 * kblo + 3] = gl::mul_pow2_f<48>(v[4 * kblo + 3], fm[kblo]);
        });
        __builtin_amdgcn_sched_barrier(RARE_FENCE_MASK);
        gl::rare_mask any = 0;
        static_for<0, 8>([&](auto I_) { any |= f0[decltype(I_)::value] | f1[decltype(I_)::value]; });
        static_for<0, 4>([&](auto I_) { any |= fm[decltype(I_)::value]; });
        if (GL_RARE_ANY(any))
            static_for<0, 4>([&](auto B_) {
                constexpr int kblo = decltype(B_)::value, b = 4 * kblo;
                constexpr bool na = NEGQ(kblo, 2), nb = NEGQ(kblo, 1) != NEGQ(kblo, 3);
                fix(bool_constant<!na>{}, integral_constant<int, 0>{}, integral_constant<int, b>{}, f0[2 * kblo], 0);
                fix(bool_constant<na>{}, integral_constant<int, 0>{}, integral_constant<int, b + 2>{}, f1[2 * kblo], 0);
                fix(bool_constant<!nb>{}, integral_constant<int, 0>{}, integral_constant<int, b + 1>{}, f0[2 * kblo + 1], 0);
                fix(bool_constant<nb>{}, integral_constant<int, 48>{}, integral_constant<int, b + 3>{}, f1[2 * kblo + 1], fm[kblo]);
            });
    }
    {   // second stage: pairs (0, 1) and (2, 3); their subtrahends carry the sign of x1
        gl::rare_mask f0[8], f1[8];
        static_for<0, 4>([&](auto B_) {
            constexpr int kblo = decltype(B_)::value, b = 4 * kblo;
            bfly(bool_constant<NEGQ(kblo, 1)>{}, integral_constant<int, b>{}, integral_constant<int, b + 1>{}, f0[2 * kblo], f1[2 * kblo]);
            bfly(bool_constant<NEGQ(kblo, 1)>{}, integral_constant<int, b + 2>{}, integral_constant<int, b + 3>{}, f0[2 * kblo + 1], f1[2 * kblo + 1]);
        });
        __builtin_amdgcn_sched_barrier(RARE_FENCE_MASK);
        gl::rare_mask any = 0;
        static_for<0, 8>([&](auto I_) { any |= f0[decltype(I_)::value] | f1[decltype(I_)::value]; });
        if (GL_RARE_ANY(any))
            static_for<0, 4>([&](auto B_) {
                constexpr int kblo = decltype(B_)::value, b = 4 * kblo;
                constexpr bool n1 = NEGQ(kblo, 1);
                fix(bool_constant<!n1>{}, integral_constant<int, 0>{}, integral_constant<int, b>{}, f0[2 * kblo], 0);
                fix(bool_constant<n1>{}, integral_constant<int, 0>{}, integral_constant<int, b + 1>{}, f1[2 * kblo], 0);
                fix(bool_constant<!n1>{}, integral_constant<int, 0>{}, integral_constant<int, b + 2>{}, f0[2 * kblo + 1], 0);
                fix(bool_constant<n1>{}, integral_constant<int, 0>{}, integral_constant<int, b + 3>{}, f1[2 * kblo + 1], 0);
            });
    }
}

// Synchronisation of the NT_ threads that share a tile: a wavefront (NT_ == 64) needs only program order.
template <int NT_>
__device__ __forceinline__ void tile_sync() {
    if constexpr (NT_ == 64) {
        // One wavefront owns the tile. The LDS executes a wavefront's operations in program order, so a later
        // ds_read of another lane's slot sees the earlier ds_write without any wait; what is needed is only that the
        // compiler keeps the program order of the accesses (it must: the indices may alias) and does not schedule
        // across this point.
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    } else {
        __syncthreads();
    }
}

// Workgroup barrier that waits for this wave's LDS traffic only. __syncthreads() also drains vmcnt (hipcc puts
// s_waitcnt vmcnt(0) in front of it), which would stall on the NEXT tile's global loads that are meant to stay in
// flight across the whole transform of the current one.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }


// ---- dynamic LDS above the default limit ---------------------------------------------------------------------------------
// hipFuncAttributeMaxDynamicSharedMemorySize is a property of (function, DEVICE): remembered per device, raised once to the
// device's own limit (so that concurrent first calls set the same value), in a slot that concurrent callers may race on freely.
struct DynamicLds {
    std::atomic<uint8_t> set[64] = {};
};
inline uint32_t device_lds_limit() {
    int dev = 0, bytes = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&bytes, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess || bytes <= 0)
        return 64 * 1024;
    return (uint32_t)bytes;
}
inline hipError_t allow_dynamic_lds(DynamicLds &st, const void *fn, uint32_t lds_bytes) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::atomic<uint8_t> &slot = st.set[dev & 63];
    if (slot.load(std::memory_order_acquire)) return hipSuccess;
    const uint32_t limit = device_lds_limit();
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds_bytes > limit ? lds_bytes : limit));  // too much: HIP says so
    if (e != hipSuccess) return e;
    slot.store(1, std::memory_order_release);
    return hipSuccess;
}

// ---- direct passes (ntt_direct.hip) -------------------------------------------------------------------------------------
// Column pass of R = 2^(8 + logg) rows on tiles of 64 >> logg adjacent columns (the planner's F_WIDE geometry), logg = 0, 1, 2.
hipError_t launch_col_direct(int logg, const PassParams &p, dim3 grid, hipStream_t stream);
// The same pass as the LAST pass of a transform (F_FINAL_COL; logg = 3, natural order only): no twiddle chain, outputs canonical.
hipError_t launch_col_direct_reversed_input(const PassParams &p, dim3 grid, hipStream_t stream);
hipError_t launch_col_direct_final(const PassParams &p, dim3 grid, hipStream_t stream);
// F_COSET passes (first pass of the coset LDE): only when this says so
bool col_direct_coset_ok(int logg, dim3 grid);
// Row pass of 1024-point rows with natural-order (transposed) output on tiles of SIXTEEN rows: the planner lays the pass out
// with logt = 4 (in_sb = 16 rows, out_sb = 16); forward and inverse (index flip, row_shift).
hipError_t launch_row_natural_direct(const PassParams &p, dim3 grid, hipStream_t stream);
// Row pass of 1024-point rows IN PLACE (bit-reversed output), any tile geometry of the planner: the rows are
// src + a * in_sa + z * in_sz + row * in_t for row < t_limit, a < grid.y, z < grid.z (in_m == 1), written to the same place in dst.
hipError_t launch_row_inplace_direct(const PassParams &p, dim3 grid, hipStream_t stream);
hipError_t launch_row_inplace_direct_2048(const PassParams &p, dim3 grid, hipStream_t stream);

}  // namespace nttk
}  // namespace plonky2_hip
