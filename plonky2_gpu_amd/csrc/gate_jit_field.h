// gate_jit_field.h — field forms that only the run-time compiled gate kernels use (gate_jit.hip appends this text to gl_field.h's
// in every unit it generates): operations with a SMALL compile-time constant, which the gate programs are full of — a base-4 limb's
// range check is l (l - 3) (l (l - 3) + 2), an `EMIT` of one bit is b (b - 1) — and which the general forms of gl_field.h pay for as
// if the constant were a field element. The peephole pass of gate_jit.hip decides where they apply; tests/test_gpu_plonk.py holds the
// compiled kernels against the interpreter (which executes the programs as written) on rows made of the values where these forms
// take their rare paths.
#pragma once

namespace gl {

// a * b + K for K <= 64: the constant rides as the addend of the first multiply-add (al*bl + K <= (2^32-1)^2 + 64 < 2^64), so it
// costs what mul costs. (K is an inline constant of the VOP3 encoding; gfx950 has no VOP3 literals, hence the bound.) The rest is
// gl::mul, instruction for instruction.
template <uint32_t K>
__device__ __forceinline__ uint64_t mul_add_small(uint64_t a, uint64_t b) {
    static_assert(K <= 64, "the addend must be an inline constant");
    uint32_t al = (uint32_t)a, ah = (uint32_t)(a >> 32), bl = (uint32_t)b, bh = (uint32_t)(b >> 32);
    uint64_t r, c1;
    asm("v_mad_u64_u32 v[32:33], vcc, %2, %4, %6\n\t"         // T = al*bl + K
        "v_mad_u64_u32 v[34:35], vcc, %2, %5, 0\n\t"
        "v_mad_u64_u32 v[36:37], %1, %3, %4, v[34:35]\n\t"
        "v_mad_u64_u32 v[38:39], vcc, %3, %5, 0\n\t"
        "v_add_co_u32_e32 v33, vcc, v33, v36\n\t"
        "v_addc_co_u32_e32 v38, vcc, v38, v37, vcc\n\t"
        "v_addc_co_u32_e32 v39, vcc, 0, v39, vcc\n\t"
        "v_subb_co_u32_e64 v32, vcc, v32, v39, %1\n\t"
        "v_subbrev_co_u32_e32 v33, vcc, 0, v33, vcc\n\t"
        "s_cbranch_vccz 1f\n\t"
        "v_cndmask_b32_e64 v42, 0, -1, vcc\n\t"
        "v_sub_co_u32_e32 v32, vcc, v32, v42\n\t"
        "v_subbrev_co_u32_e32 v33, vcc, 0, v33, vcc\n\t"
        "1:\n\t"
        "v_mad_u64_u32 v[32:33], vcc, v38, -1, v[32:33]\n\t"
        "v_cndmask_b32_e64 v42, 0, -1, vcc\n\t"
        "v_mad_u64_u32 %0, vcc, v42, 1, v[32:33]"
        : "=&v"(r), "=&s"(c1)
        : "v"(al), "v"(ah), "v"(bl), "v"(bh), "n"(K)
        : "vcc", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v42");
    return r;
}

// a + K and a - K for any representative a and K < 2^32: the plain 64-bit sum / difference IS a representative unless it wraps, which
// needs a within K of 2^64 (of 0): two instructions and a wave-uniform branch over the one wrap correction (2^64 = 2^32 - 1 mod p; the
// corrected sum is below 2^33, the corrected difference at least p - K: neither can wrap again).
template <uint32_t K>
__device__ __forceinline__ uint64_t add_small(uint64_t a) {
    uint32_t al = (uint32_t)a, ah = (uint32_t)(a >> 32), rl, rh;
    asm("v_add_co_u32_e32 %0, vcc, %4, %2\n\t"
        "v_addc_co_u32_e32 %1, vcc, 0, %3, vcc\n\t"
        "s_cbranch_vccz 1f\n\t"
        "v_cndmask_b32_e64 v42, 0, -1, vcc\n\t"
        "v_add_co_u32_e32 %0, vcc, %0, v42\n\t"
        "v_addc_co_u32_e32 %1, vcc, 0, %1, vcc\n\t"
        "1:"
        : "=&v"(rl), "=&v"(rh)
        : "v"(al), "v"(ah), "s"(K)
        : "vcc", "v42");
    return ((uint64_t)rh << 32) | rl;
}

template <uint32_t K>
__device__ __forceinline__ uint64_t sub_small(uint64_t a) {
    uint32_t al = (uint32_t)a, ah = (uint32_t)(a >> 32), rl, rh;
    asm("v_subrev_co_u32_e32 %0, vcc, %4, %2\n\t"             // a.lo - K
        "v_subbrev_co_u32_e32 %1, vcc, 0, %3, vcc\n\t"
        "s_cbranch_vccz 1f\n\t"
        "v_cndmask_b32_e64 v42, 0, -1, vcc\n\t"
        "v_sub_co_u32_e32 %0, vcc, %0, v42\n\t"
        "v_subbrev_co_u32_e32 %1, vcc, 0, %1, vcc\n\t"
        "1:"
        : "=&v"(rl), "=&v"(rh)
        : "v"(al), "v"(ah), "s"(K)
        : "vcc", "v42");
    return ((uint64_t)rh << 32) | rl;
}

// a * K and a + K for a 64-bit compile-time constant (a gate's own constants: round constants and matrix entries of PoseidonGate):
// gl::mul / gl::add with the constant's halves as SCALAR operands — every multiply-add and add of those forms reads at most one
// half of b — so that no vector register pair has to be loaded with it first (two v_mov per LOAD_IMM).
template <uint64_t K>
__device__ __forceinline__ uint64_t mul_k(uint64_t a) {
    uint32_t al = (uint32_t)a, ah = (uint32_t)(a >> 32);
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
        "s_cbranch_vccz 1f\n\t"
        "v_cndmask_b32_e64 v42, 0, -1, vcc\n\t"
        "v_sub_co_u32_e32 v32, vcc, v32, v42\n\t"
        "v_subbrev_co_u32_e32 v33, vcc, 0, v33, vcc\n\t"
        "1:\n\t"
        "v_mad_u64_u32 v[32:33], vcc, v38, -1, v[32:33]\n\t"
        "v_cndmask_b32_e64 v42, 0, -1, vcc\n\t"
        "v_mad_u64_u32 %0, vcc, v42, 1, v[32:33]"
        : "=&v"(r), "=&s"(c1)
        : "v"(al), "v"(ah), "s"((uint32_t)K), "s"((uint32_t)(K >> 32))
        : "vcc", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v42");
    return r;
}

// (An add-with-carry cannot read a scalar register next to VCC — one scalar source per instruction on gfx950 — so the sum is formed
// as a.lo * 1 + K, which cannot wrap for a canonical K = p - 1 = 2^64 - 2^32 at most, and a.hi is added to its upper half: the carry
// of that add is the carry of the 64-bit sum. Four instructions like gl::add.)
template <uint64_t K>
__device__ __forceinline__ uint64_t add_k(uint64_t a) {
    static_assert(K < 0xFFFFFFFF00000001ull, "the constant must be canonical");
    uint32_t al = (uint32_t)a, ah = (uint32_t)(a >> 32);
    uint64_t r;
    asm("v_mad_u64_u32 v[32:33], vcc, %1, 1, %3\n\t"
        "v_add_co_u32_e32 v33, vcc, %2, v33\n\t"
        "v_cndmask_b32_e64 v42, 0, -1, vcc\n\t"
        "v_mad_u64_u32 %0, vcc, v42, 1, v[32:33]\n\t"
        "s_cbranch_vccz 1f\n\t"
        "v_cndmask_b32_e64 v42, 0, -1, vcc\n\t"
        "v_mad_u64_u32 %0, vcc, v42, 1, %0\n\t"
        "1:"
        : "=&v"(r)
        : "v"(al), "v"(ah), "s"(K)
        : "vcc", "v32", "v33", "v42");
    return r;
}

// sum_k c_k * alpha^k with TWO column accumulators instead of gl::DotAcc's three: with beta_k = alpha^k * 2^32 (mod p) from the
// same table, c * alpha^k = c.lo * alpha^k + c.hi * beta_k, and both products have only the columns 2^0 and 2^32:
//     A0 += c.lo * alpha.lo + c.hi * beta.lo        A1 += c.lo * alpha.hi + c.hi * beta.hi
// The same four multiply-adds and four carry counts per term as gl::dot_term, but six registers per challenge where DotAcc takes nine
// — and in a fused unit the accumulators of all its gates are live from the first statement to the last.
struct DotCol2 {
    uint64_t a0 = 0, a1 = 0;
    uint32_t k0 = 0, k1 = 0;
};

__device__ __forceinline__ void dot_term2(DotCol2 &d, uint64_t c, uint64_t alpha, uint64_t beta) {
    uint32_t cl = (uint32_t)c, ch = (uint32_t)(c >> 32);
    uint32_t al = (uint32_t)alpha, ah = (uint32_t)(alpha >> 32), bl = (uint32_t)beta, bh = (uint32_t)(beta >> 32);
    uint64_t c0, c1, c2, c3;
    // a carry written to a scalar pair is read three instructions later (two wait states are needed, see gl::dot_term)
    asm("v_mad_u64_u32 %0, %4, %8, %10, %0\n\t"
        "v_mad_u64_u32 %1, %5, %8, %11, %1\n\t"
        "v_mad_u64_u32 %0, %6, %9, %12, %0\n\t"
        "v_mad_u64_u32 %1, %7, %9, %13, %1\n\t"
        "v_addc_co_u32_e64 %2, vcc, 0, %2, %4\n\t"
        "v_addc_co_u32_e64 %3, vcc, 0, %3, %5\n\t"
        "v_addc_co_u32_e64 %2, vcc, 0, %2, %6\n\t"
        "v_addc_co_u32_e64 %3, vcc, 0, %3, %7"
        : "+v"(d.a0), "+v"(d.a1), "+v"(d.k0), "+v"(d.k1), "=&s"(c0), "=&s"(c1), "=&s"(c2), "=&s"(c3)
        : "v"(cl), "v"(ch), "s"(al), "s"(ah), "s"(bl), "s"(bh)
        : "vcc");
}

__device__ __forceinline__ uint64_t dot_finish2(const DotCol2 &d) {
    DotAcc full;
    full.a0 = d.a0, full.a1 = d.a1, full.a2 = 0, full.k0 = d.k0, full.k1 = d.k1, full.k2 = 0;
    return dot_finish(full);
}

}  // namespace gl
