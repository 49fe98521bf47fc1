// gl_field.cuh — Goldilocks field arithmetic for gfx950 device code.
//
// p = 2^64 - 2^32 + 1. Values are plain-domain u64; like the reference
// (field/src/goldilocks_field.rs:26) every u64 is a legal representative inside a kernel and
// results are canonicalised (gl_canon) only when they are stored to a boundary buffer.
// The 128->64 reduction is the special-form one of goldilocks_field.rs:345-358
// (lo - hi_hi + hi_lo*(2^32-1)), not Montgomery: it keeps values in the plain domain so nothing
// has to be converted at the C-ABI.
//
// CDNA4 has no 64-bit integer multiplier: a 64x64->128 product is four v_mad_u64_u32
// (32x32+64->64). 64-bit adds are single v_lshl_add_u64 instructions. Everything here is
// branch-free (v_cndmask) — a wavefront cannot profit from the "rare branch" the CPU code uses.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gl {

typedef unsigned __int128 u128;

static constexpr uint64_t P = 0xFFFFFFFF00000001ULL;
static constexpr uint64_t EPS = 0xFFFFFFFFULL;  // 2^64 mod p

__device__ __forceinline__ uint64_t canon(uint64_t a) { return a >= P ? a - P : a; }

// a + b for arbitrary representatives (goldilocks_field.rs:197-219).
__device__ __forceinline__ uint64_t add(uint64_t a, uint64_t b) {
    uint64_t s = a + b;
    uint64_t s1 = s + ((s < a) ? EPS : 0);
    return s1 + ((s1 < s) ? EPS : 0);
}

// a + b where b is canonical (< p): a single wrap correction suffices.
__device__ __forceinline__ uint64_t add_canonical(uint64_t a, uint64_t b) {
    uint64_t s = a + b;
    return s + ((s < a) ? EPS : 0);
}

// a - b for arbitrary representatives (goldilocks_field.rs:234-256).
__device__ __forceinline__ uint64_t sub(uint64_t a, uint64_t b) {
    uint64_t d = a - b;
    uint64_t d1 = d - ((a < b) ? EPS : 0);
    return d1 - ((d1 > d) ? EPS : 0);
}

__device__ __forceinline__ uint64_t neg(uint64_t a) {
    uint64_t c = canon(a);
    return c ? P - c : 0;
}

// x = lo + 2^64*hi  ->  representative in [0, 2^64) (goldilocks_field.rs:345-358).
__device__ __forceinline__ uint64_t reduce128(uint64_t lo, uint64_t hi) {
    uint64_t hh = hi >> 32, hl = hi & EPS;
    uint64_t t0 = lo - hh;
    t0 -= (lo < hh) ? EPS : 0;
    uint64_t t1 = (hl << 32) - hl;  // hl * (2^32 - 1), shift/sub instead of a fifth multiply
    uint64_t r = t0 + t1;
    return r + ((r < t0) ? EPS : 0);
}

// x = lo + 2^64*hi with hi < 2^32 (sums of <=2^32 products of u64 by small constants).
__device__ __forceinline__ uint64_t reduce96(uint64_t lo, uint32_t hi) {
    uint64_t t1 = ((uint64_t)hi << 32) - hi;
    uint64_t r = lo + t1;
    return r + ((r < lo) ? EPS : 0);
}

__device__ __forceinline__ void mul_wide(uint64_t a, uint64_t b, uint64_t &lo, uint64_t &hi) {
    u128 x = (u128)a * (u128)b;
    lo = (uint64_t)x;
    hi = (uint64_t)(x >> 64);
}

__device__ __forceinline__ uint64_t mul(uint64_t a, uint64_t b) {
    uint64_t lo, hi;
    mul_wide(a, b, lo, hi);
    return reduce128(lo, hi);
}

__device__ __forceinline__ uint64_t sqr(uint64_t a) { return mul(a, a); }

// acc + x*y (goldilocks_field.rs:119-123); u64 + u64*u64 cannot overflow 128 bits.
__device__ __forceinline__ uint64_t mac(uint64_t acc, uint64_t x, uint64_t y) {
    u128 t = (u128)x * (u128)y + (u128)acc;
    return reduce128((uint64_t)t, (uint64_t)(t >> 64));
}

// x * 2^k mod p for a compile-time 0 <= k < 192, multiply-free.
// 2^96 = -1 (mod p) so k >= 96 is a negated shift by k-96; for k < 96 the 160-bit value
// x*2^k = lo + mid*2^64 + top*2^96 reduces to lo - top + mid*(2^32-1).
template <int K>
__device__ __forceinline__ uint64_t mul_pow2(uint64_t x) {
    static_assert(K >= 0 && K < 192, "shift out of range");
    if constexpr (K == 0) {
        return x;
    } else if constexpr (K >= 96) {
        return neg(mul_pow2<K - 96>(x));
    } else if constexpr (K < 32) {
        uint64_t lo = x << K, hi = x >> (64 - K);  // hi < 2^32
        return reduce96(lo, (uint32_t)hi);
    } else if constexpr (K == 32) {
        return reduce128(x << 32, x >> 32);
    } else if constexpr (K < 64) {
        return reduce128(x << K, x >> (64 - K));
    } else if constexpr (K == 64) {
        return reduce128(0, x);
    } else {
        // 64 < K < 96: value = (x << (K-64)) * 2^64 ; the part above 2^96 is top = x >> (160-K)...
        // split x*2^(K-64) = h (128-bit: hlo + hhi*2^64), then x*2^K = hlo*2^64 + hhi*2^128,
        // and 2^128 = -2^32 (mod p): result = reduce128(0, hlo) - hhi*2^32.
        constexpr int S = K - 64;  // 1..31
        uint64_t hlo = x << S, hhi = x >> (64 - S);  // hhi < 2^31
        return sub(reduce128(0, hlo), hhi << 32);
    }
}

__device__ __forceinline__ uint64_t pow(uint64_t base, uint64_t e) {
    uint64_t cur = base, acc = 1;
    while (e) {
        if (e & 1) acc = mul(acc, cur);
        cur = sqr(cur);
        e >>= 1;
    }
    return acc;
}

// x^7 (plonky2/src/hash/poseidon.rs:522-528)
__device__ __forceinline__ uint64_t pow7(uint64_t x) {
    uint64_t x2 = sqr(x), x4 = sqr(x2), x3 = mul(x, x2);
    return mul(x3, x4);
}

}  // namespace gl

// ---- host-side twins (table construction, n_inv, ...) -------------------------------------
namespace glh {
typedef unsigned __int128 u128;
static constexpr uint64_t P = 0xFFFFFFFF00000001ULL;
inline uint64_t mul(uint64_t a, uint64_t b) { return (uint64_t)(((u128)(a % P) * (u128)(b % P)) % P); }
inline uint64_t add(uint64_t a, uint64_t b) { return (uint64_t)(((u128)(a % P) + (b % P)) % P); }
inline uint64_t pow(uint64_t b, uint64_t e) {
    uint64_t acc = 1, cur = b % P;
    while (e) {
        if (e & 1) acc = mul(acc, cur);
        cur = mul(cur, cur);
        e >>= 1;
    }
    return acc;
}
inline uint64_t inv(uint64_t a) { return pow(a, P - 2); }
// Field::primitive_root_of_unity (field/src/types.rs:268-272)
inline uint64_t root_of_unity(unsigned n_log) { return pow(1753635133440165772ULL, 1ULL << (32 - n_log)); }
}  // namespace glh
