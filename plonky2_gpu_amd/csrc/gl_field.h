// gl_field.h — Goldilocks field arithmetic for gfx950 device code.
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
#ifdef GL_JIT
// compiled at run time by hiprtc (gate_jit.hip): no system headers there, the HIP builtins are implicit
typedef unsigned long long uint64_t;
typedef long long int64_t;
typedef unsigned int uint32_t;
typedef int int32_t;
typedef unsigned short uint16_t;
#else
#include <hip/hip_runtime.h>
#include <stdint.h>
#endif

namespace gl {

typedef unsigned __int128 u128;

static constexpr uint64_t P = 0xFFFFFFFF00000001ULL;
static constexpr uint64_t EPS = 0xFFFFFFFFULL;  // 2^64 mod p

__device__ __forceinline__ uint64_t canon(uint64_t a) { return a >= P ? a - P : a; }

// a + b for arbitrary representatives (goldilocks_field.rs:197-219): the portable statement of what
// gl::add (carry-flag version, below) computes.
__device__ __forceinline__ uint64_t add_generic(uint64_t a, uint64_t b) {
    uint64_t s = a + b;
    uint64_t s1 = s + ((s < a) ? EPS : 0);
    return s1 + ((s1 < s) ? EPS : 0);
}

// a + b where b is canonical (< p): a single wrap correction suffices.
__device__ __forceinline__ uint64_t add_canonical(uint64_t a, uint64_t b) {
    uint64_t s = a + b;
    return s + ((s < a) ? EPS : 0);
}

// a - b for arbitrary representatives (goldilocks_field.rs:234-256): the portable statement of gl::sub.
__device__ __forceinline__ uint64_t sub_generic(uint64_t a, uint64_t b) {
    uint64_t d = a - b;
    uint64_t d1 = d - ((a < b) ? EPS : 0);
    return d1 - ((d1 > d) ? EPS : 0);
}

// The same two operations in hand-scheduled form. What costs on gfx950 is not the instruction count but the
// carry-flag instructions: v_add_co / v_addc_co / v_sub_co / v_subb_co issue at ~5 cycles per wavefront, as much
// as a v_mad_u64_u32 (~6) and twice a plain v_mov / v_xor (2.5); a flag-free 64-bit v_lshl_add_u64 is 4.4
// (tools/ubench_issue.hip). So:
//   - the FIRST wrap correction of a sum is one multiply-add, r = t*1 + s with t = carry ? 2^32-1 : 0, which also
//     delivers the carry-out that says whether a second correction is needed;
//   - the SECOND correction (the reference's double wrap, goldilocks_field.rs:203-256) can only trigger when both
//     operands are >= p, i.e. non-canonical representatives from a window of 2^32 values: it sits behind a
//     wave-uniform s_cbranch_vccz — a branch that is essentially never taken costs a scalar instruction, not a
//     divergent wavefront. (A data-dependent branch that IS sometimes taken would not pay; this one does.)
// add: 4 VALU (was 8), 26.5 cycles per wavefront (was 37.7); sub: 5 VALU, 26.6 (was 36.2) — tools/ubench_field.hip,
// which also checks both against big integers on 4.6 M operand pairs including 200 000 with both operands >= p and
// single-wavefront launches. VCC written by a VALU instruction and read by the next (carry-in, v_cndmask mask,
// s_cbranch_vccz) needs no wait states on gfx950: measured, see mul below.
__device__ __forceinline__ uint64_t add(uint64_t a, uint64_t b) {
    uint32_t al = (uint32_t)a, ah = (uint32_t)(a >> 32), bl = (uint32_t)b, bh = (uint32_t)(b >> 32);
    uint64_t r;
    asm("v_add_co_u32_e32 v32, vcc, %1, %3\n\t"
        "v_addc_co_u32_e32 v33, vcc, %2, %4, vcc\n\t"
        "v_cndmask_b32_e64 v42, 0, -1, vcc\n\t"                  // overflow: 2^64 = 2^32 - 1
        "v_mad_u64_u32 %0, vcc, v42, 1, v[32:33]\n\t"          // r = s + t, carry -> vcc
        "s_cbranch_vccz 1f\n\t"
        "v_cndmask_b32_e64 v42, 0, -1, vcc\n\t"                  // rare: once more (cannot overflow a third time)
        "v_mad_u64_u32 %0, vcc, v42, 1, %0\n\t"
        "1:"
        : "=&v"(r)
        : "v"(al), "v"(ah), "v"(bl), "v"(bh)
        : "vcc", "v32", "v33", "v42");
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

// ---------------------------------------------------------------------------------------------
// Canonical-domain primitives (inputs and outputs < p), hand-scheduled carry chains.
//
// hipcc lowers `s < a` carry tests to v_cmp_lt_u64 + v_lshl_add_u64 (both double-pumped 64-bit
// ops) and never uses the carry-out of v_add_co/v_addc, so the portable versions above cost
// 15-21 lane-cycles per add/sub (measured, profiles/r01_v1_ubench.txt). These use the carry
// flag directly: 5-7 single-rate VALU instructions. `s_nop 1` = the two wait states gfx950 needs
// between a VALU instruction that writes VCC/SGPR and a VALU instruction that reads it as
// carry-in or select mask (the compiler inserts the same for its own code; inside asm we must).
// ---------------------------------------------------------------------------------------------

__device__ __forceinline__ uint64_t pack64(uint32_t lo, uint32_t hi) { return ((uint64_t)hi << 32) | lo; }

// x (any u64) -> x mod p
__device__ __forceinline__ uint64_t canon_c(uint64_t x) {
    uint32_t xl = (uint32_t)x, xh = (uint32_t)(x >> 32), rl, rh;
    asm("v_add_co_u32_e32 %0, vcc, -1, %2\n\t"       // t = x + (2^32-1): carries out iff x >= p
        "s_nop 1\n\t"
        "v_addc_co_u32_e32 %1, vcc, 0, %3, vcc\n\t"
        "s_nop 1\n\t"
        "v_cndmask_b32_e32 %0, %2, %0, vcc\n\t"
        "v_cndmask_b32_e32 %1, %3, %1, vcc"
        : "=&v"(rl), "=&v"(rh)
        : "v"(xl), "v"(xh)
        : "vcc");
    return pack64(rl, rh);
}

// a, b < p  ->  (a + b) mod p, canonical
__device__ __forceinline__ uint64_t add_c(uint64_t a, uint64_t b) {
    uint32_t al = (uint32_t)a, ah = (uint32_t)(a >> 32), bl = (uint32_t)b, bh = (uint32_t)(b >> 32);
    uint32_t sl, sh, tl, th;
    uint64_t c1;
    asm("v_add_co_u32_e32 %0, vcc, %5, %7\n\t"       // s = a + b, carry c1
        "s_nop 1\n\t"
        "v_addc_co_u32_e64 %1, %4, %6, %8, vcc\n\t"
        "v_add_co_u32_e32 %2, vcc, -1, %0\n\t"       // t = s - p (mod 2^64), carry c2 iff s >= p
        "s_nop 1\n\t"
        "v_addc_co_u32_e32 %3, vcc, 0, %1, vcc\n\t"
        "s_or_b64 vcc, vcc, %4\n\t"                   // take t when the true sum was >= p
        "s_nop 1\n\t"
        "v_cndmask_b32_e32 %0, %0, %2, vcc\n\t"
        "v_cndmask_b32_e32 %1, %1, %3, vcc"
        : "=&v"(sl), "=&v"(sh), "=&v"(tl), "=&v"(th), "=&s"(c1)
        : "v"(al), "v"(ah), "v"(bl), "v"(bh)
        : "vcc");
    return pack64(sl, sh);
}

// a, b < p  ->  (a - b) mod p, canonical
__device__ __forceinline__ uint64_t sub_c(uint64_t a, uint64_t b) {
    uint32_t al = (uint32_t)a, ah = (uint32_t)(a >> 32), bl = (uint32_t)b, bh = (uint32_t)(b >> 32);
    uint32_t dl, dh, e;
    asm("v_sub_co_u32_e32 %0, vcc, %3, %5\n\t"       // d = a - b, borrow
        "s_nop 1\n\t"
        "v_subb_co_u32_e32 %1, vcc, %4, %6, vcc\n\t"
        "s_nop 1\n\t"
        "v_cndmask_b32_e64 %2, 0, -1, vcc\n\t"        // e = borrow ? 2^32-1 : 0
        "v_sub_co_u32_e32 %0, vcc, %0, %2\n\t"        // d += p  ==  d -= (2^32-1)  (mod 2^64)
        "s_nop 1\n\t"
        "v_subbrev_co_u32_e32 %1, vcc, 0, %1, vcc"
        : "=&v"(dl), "=&v"(dh), "=&v"(e)
        : "v"(al), "v"(ah), "v"(bl), "v"(bh)
        : "vcc");
    return pack64(dl, dh);
}

// x = lo + 2^64*hi (any 128-bit value)  ->  x mod p, canonical.
//   x = lo - hh + hl*(2^32-1)  with hi = hh*2^32 + hl   (goldilocks_field.rs:345-358)
__device__ __forceinline__ uint64_t reduce128_c(uint64_t lo, uint64_t hi) {
    uint32_t ll = (uint32_t)lo, lh = (uint32_t)(lo >> 32), hl = (uint32_t)hi, hh = (uint32_t)(hi >> 32);
    uint32_t rl, rh, ul, uh, e;
    asm("v_sub_co_u32_e32 %0, vcc, %5, %8\n\t"       // t0 = lo - hh
        "s_nop 1\n\t"
        "v_subbrev_co_u32_e32 %1, vcc, 0, %6, vcc\n\t"
        "s_nop 1\n\t"
        "v_cndmask_b32_e64 %4, 0, -1, vcc\n\t"        // borrow: t0 -= 2^32-1
        "v_sub_co_u32_e32 %0, vcc, %0, %4\n\t"
        "s_nop 1\n\t"
        "v_subbrev_co_u32_e32 %1, vcc, 0, %1, vcc\n\t"
        "v_sub_co_u32_e32 %2, vcc, 0, %7\n\t"         // u = hl*(2^32-1) = (hl<<32) - hl
        "s_nop 1\n\t"
        "v_subbrev_co_u32_e32 %3, vcc, 0, %7, vcc\n\t"
        "v_add_co_u32_e32 %0, vcc, %0, %2\n\t"        // r = t0 + u
        "s_nop 1\n\t"
        "v_addc_co_u32_e32 %1, vcc, %1, %3, vcc\n\t"
        "s_nop 1\n\t"
        "v_cndmask_b32_e64 %4, 0, -1, vcc\n\t"        // carry: r += 2^32-1 (cannot carry again)
        "v_add_co_u32_e32 %0, vcc, %0, %4\n\t"
        "s_nop 1\n\t"
        "v_addc_co_u32_e32 %1, vcc, 0, %1, vcc\n\t"
        "v_add_co_u32_e32 %2, vcc, -1, %0\n\t"        // canonicalise: r >= p ? r - p : r
        "s_nop 1\n\t"
        "v_addc_co_u32_e32 %3, vcc, 0, %1, vcc\n\t"
        "s_nop 1\n\t"
        "v_cndmask_b32_e32 %0, %0, %2, vcc\n\t"
        "v_cndmask_b32_e32 %1, %1, %3, vcc"
        : "=&v"(rl), "=&v"(rh), "=&v"(ul), "=&v"(uh), "=&v"(e)
        : "v"(ll), "v"(lh), "v"(hl), "v"(hh)
        : "vcc");
    return pack64(rl, rh);
}

__device__ __forceinline__ void mul_wide(uint64_t a, uint64_t b, uint64_t &lo, uint64_t &hi) {
    u128 x = (u128)a * (u128)b;
    lo = (uint64_t)x;
    hi = (uint64_t)(x >> 64);
}

// a * b mod p for arbitrary representatives, result in [0, 2^64): hand-scheduled.
//
// hipcc's lowering of (u128)a*b + reduce128 spends ~24 VALU instructions plus hazard padding: six
// v_mov to build zero-extended register pairs for the v_mad_u64_u32 addends and compare/select
// corrections built from double-pumped 64-bit ops. Here (16 VALU):
//   product:  T = al*bl ; U = al*bh ; V = ah*bl + U (carry c1 into an SGPR pair) ; W = ah*bh
//             -> lo = (T.lo, T.hi + V.lo), hl = W.lo + V.hi + carry, hh = W.hi + carry   (three chained adds)
//             No multiply-add needs a constructed addend: the only one is U, a pair a previous
//             v_mad_u64_u32 wrote (a zero-extended half would cost a v_mov each, four in the first version).
//             c1 has weight 2^96 = -1 (mod p), the weight of hh: it is the BORROW-IN of the subtraction below.
//   reduce :  t0 = lo - hh - c1 ; r = t0 + hl*(2^32-1) as ONE v_mad_u64_u32 whose carry-out drives the last
//             correction, itself one multiply-add (t*1 + r) writing the result pair (goldilocks_field.rs:345-358).
//             The borrow of t0 (lo < hh + c1: probability ~2^-32 on real data, certain for e.g. 2^63 * 2^63) is
//             corrected behind a wave-uniform branch, like the second correction of add/sub.
//   12 VALU + a scalar branch, 57.9 cycles per wavefront (16 VALU: 72.7; the first version, 19 VALU: ~77).
// No wait states are spent between a VALU instruction that writes VCC and the next one that reads it
// (carry-in, or v_cndmask's mask): gfx950 interlocks VCC. Measured, not assumed: the schedule with and
// without `s_nop 1` pads agrees on 4M random + all edge-operand pairs, also when a single wavefront
// runs alone and issues back to back; explicit SGPR-pair carries (dot_term) DO need two wait states.
__device__ __forceinline__ uint64_t mul(uint64_t a, uint64_t b) {
    uint32_t al = (uint32_t)a, ah = (uint32_t)(a >> 32), bl = (uint32_t)b, bh = (uint32_t)(b >> 32);
    uint64_t r, c1;
    // LLVM's AMDGPU inline asm has no sub-register operand modifier, so the 64-bit temporaries
    // whose halves are needed live in fixed registers v[32:42] (declared clobbered). LOW registers on purpose:
    // a kernel's VGPR allocation is the highest register it touches, and with these at v116-v126 (where they
    // first were) every kernel reported 127 VGPRs = 4 waves per SIMD whatever it needed; at v32 the NTT passes
    // take 75-79 (6 waves) and the tree-layer kernel 88 (5). v32+ is above the argument registers of the
    // device-function calling convention (v0-v31), so the run-time compiled gate functions are unaffected.
    asm("v_mad_u64_u32 v[32:33], vcc, %2, %4, 0\n\t"          // T = al*bl
        "v_mad_u64_u32 v[34:35], vcc, %2, %5, 0\n\t"          // U = al*bh
        "v_mad_u64_u32 v[36:37], %1, %3, %4, v[34:35]\n\t"  // V = ah*bl + U, carry c1 (weight 2^96)
        "v_mad_u64_u32 v[38:39], vcc, %3, %5, 0\n\t"          // W = ah*bh
        "v_add_co_u32_e32 v33, vcc, v33, v36\n\t"            // lo.hi = T.hi + V.lo        lo = (v32, v33)
        "v_addc_co_u32_e32 v38, vcc, v38, v37, vcc\n\t"      // hl = W.lo + V.hi + carry
        "v_addc_co_u32_e32 v39, vcc, 0, v39, vcc\n\t"         // hh = W.hi + carry (+ c1, applied next)
        "v_subb_co_u32_e64 v32, vcc, v32, v39, %1\n\t"       // t0 = lo - hh - c1
        "v_subbrev_co_u32_e32 v33, vcc, 0, v33, vcc\n\t"
        "s_cbranch_vccz 1f\n\t"
        "v_cndmask_b32_e64 v42, 0, -1, vcc\n\t"                // rare: borrow => t0 -= 2^32-1 (cannot borrow again)
        "v_sub_co_u32_e32 v32, vcc, v32, v42\n\t"
        "v_subbrev_co_u32_e32 v33, vcc, 0, v33, vcc\n\t"
        "1:\n\t"
        "v_mad_u64_u32 v[32:33], vcc, v38, -1, v[32:33]\n\t"  // r = t0 + hl*(2^32-1), carry -> vcc
        "v_cndmask_b32_e64 v42, 0, -1, vcc\n\t"                // carry: r += 2^32-1 (cannot carry again)
        "v_mad_u64_u32 %0, vcc, v42, 1, v[32:33]"
        : "=&v"(r), "=&s"(c1)
        : "v"(al), "v"(ah), "v"(bl), "v"(bh)
        : "vcc", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v42");
    return r;
}

__device__ __forceinline__ uint64_t sqr(uint64_t a) { return mul(a, a); }

// ---- the same three operations with their RARE PATH DEFERRED (round 4) -----------------------------------------------------
// add, sub and mul above each end their fast path with `s_cbranch_vccz` around a correction that almost never runs (add / sub: both
// operands >= p; mul: lo < hh + c1, probability 2^-32 per lane). The branch itself is what costs: the wave cannot issue anything
// until the vector instruction that wrote VCC has left the pipeline, and a pass whose waves stall together (the NTT tiles: barriers,
// LDS round trips) cannot hide it: the direct passes run 7 % (natural order) to 11 % (in place) faster with the branches removed, the
// plain Poseidon kernel 13 % faster at two waves per SIMD and no faster at four (profiles/r04_rare_path_branches.jsonl).
// The *_f variants run the fast path only and hand back, in a scalar register pair, the mask of lanes whose rare condition fired
// (the carry- / borrow-out that the branch tested); the caller ORs the masks of a GROUP of independent operations, branches ONCE,
// and applies *_fix to the flagged results. Every correction can be applied after the fact:
//   add: the second wrap adds 2^64 = e once more; the wrapped sum is below e, so r + e cannot wrap again
//   sub: the second borrow subtracts e once more; the wrapped difference is above 2^64 - e, so r - e cannot borrow again
//   mul: the fast path reduces (t0 + 2^64) instead of t0, exactly (its own wrap correction is exact for what it is given):
//        r_fast = r + e (mod p), so the true product is sub(r_fast, e)
// Exactness is that of add / sub / mul: tools/ubench_field.hip and tests/test_gpu_field.py run the grouped forms over the same edge
// operands (both >= p, 2^63 * 2^63, ...), with the flagged operation first, last and alone in its group.
typedef uint64_t rare_mask;  // lanes whose result needs its correction; lives in an SGPR pair

__device__ __forceinline__ uint64_t add_f(uint64_t a, uint64_t b, rare_mask &f) {
    uint32_t al = (uint32_t)a, ah = (uint32_t)(a >> 32), bl = (uint32_t)b, bh = (uint32_t)(b >> 32);
    uint64_t r;
    asm("v_add_co_u32_e32 v32, vcc, %2, %4\n\t"
        "v_addc_co_u32_e32 v33, vcc, %3, %5, vcc\n\t"
        "v_cndmask_b32_e64 v42, 0, -1, vcc\n\t"                  // overflow: 2^64 = 2^32 - 1
        "v_mad_u64_u32 %0, %1, v42, 1, v[32:33]"                  // r = s + t, carry -> the mask
        : "=&v"(r), "=&s"(f)
        : "v"(al), "v"(ah), "v"(bl), "v"(bh)
        : "vcc", "v32", "v33", "v42");
    return r;
}
__device__ __forceinline__ uint64_t add_fix(uint64_t r, rare_mask f) {
    asm volatile("v_cndmask_b32_e64 v42, 0, -1, %1\n\t"
                 "v_mad_u64_u32 %0, vcc, v42, 1, %0"
                 : "+v"(r)
                 : "s"(f)
                 : "vcc", "v42");
    return r;
}

__device__ __forceinline__ uint64_t sub_f(uint64_t a, uint64_t b, rare_mask &f) {
    uint32_t al = (uint32_t)a, ah = (uint32_t)(a >> 32), bl = (uint32_t)b, bh = (uint32_t)(b >> 32);
    uint32_t rl, rh, t;
    asm("v_sub_co_u32_e32 %0, vcc, %4, %6\n\t"
        "v_subb_co_u32_e32 %1, vcc, %5, %7, vcc\n\t"
        "v_cndmask_b32_e64 %2, 0, -1, vcc\n\t"                    // borrow: -2^64 = -(2^32 - 1)
        "v_sub_co_u32_e32 %0, vcc, %0, %2\n\t"
        "v_subbrev_co_u32_e64 %1, %3, 0, %1, vcc"                  // borrow -> the mask
        : "=&v"(rl), "=&v"(rh), "=&v"(t), "=&s"(f)
        : "v"(al), "v"(ah), "v"(bl), "v"(bh)
        : "vcc");
    return ((uint64_t)rh << 32) | rl;
}
__device__ __forceinline__ uint64_t sub_fix(uint64_t r, rare_mask f) {
    uint32_t rl = (uint32_t)r, rh = (uint32_t)(r >> 32), t;
    asm volatile("v_cndmask_b32_e64 %2, 0, -1, %3\n\t"
                 "v_sub_co_u32_e32 %0, vcc, %0, %2\n\t"
                 "v_subbrev_co_u32_e32 %1, vcc, 0, %1, vcc"
                 : "+v"(rl), "+v"(rh), "=&v"(t)
                 : "s"(f)
                 : "vcc");
    return ((uint64_t)rh << 32) | rl;
}

__device__ __forceinline__ uint64_t mul_f(uint64_t a, uint64_t b, rare_mask &f) {
    uint32_t al = (uint32_t)a, ah = (uint32_t)(a >> 32), bl = (uint32_t)b, bh = (uint32_t)(b >> 32);
    uint64_t r, c1;
    asm("v_mad_u64_u32 v[32:33], vcc, %3, %5, 0\n\t"          // T = al*bl
        "v_mad_u64_u32 v[34:35], vcc, %3, %6, 0\n\t"          // U = al*bh
        "v_mad_u64_u32 v[36:37], %1, %4, %5, v[34:35]\n\t"    // V = ah*bl + U, carry c1 (weight 2^96)
        "v_mad_u64_u32 v[38:39], vcc, %4, %6, 0\n\t"          // W = ah*bh
        "v_add_co_u32_e32 v33, vcc, v33, v36\n\t"            // lo.hi = T.hi + V.lo        lo = (v32, v33)
        "v_addc_co_u32_e32 v38, vcc, v38, v37, vcc\n\t"      // hl = W.lo + V.hi + carry
        "v_addc_co_u32_e32 v39, vcc, 0, v39, vcc\n\t"         // hh = W.hi + carry (+ c1, applied next)
        "v_subb_co_u32_e64 v32, vcc, v32, v39, %1\n\t"       // t0 = lo - hh - c1
        "v_subbrev_co_u32_e64 v33, %2, 0, v33, vcc\n\t"      // its borrow -> the mask (the fast path goes on with t0 + 2^64)
        "v_mad_u64_u32 v[32:33], vcc, v38, -1, v[32:33]\n\t"  // r = t0 + hl*(2^32-1), carry -> vcc
        "v_cndmask_b32_e64 v42, 0, -1, vcc\n\t"                // carry: r += 2^32-1 (cannot carry again)
        "v_mad_u64_u32 %0, vcc, v42, 1, v[32:33]"
        : "=&v"(r), "=&s"(c1), "=&s"(f)
        : "v"(al), "v"(ah), "v"(bl), "v"(bh)
        : "vcc", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v42");
    return r;
}
// r_fast = r + e (mod p) in the flagged lanes: subtract e there, with sub's own (exact, two-borrow) arithmetic
__device__ __forceinline__ uint64_t mul_fix(uint64_t r, rare_mask f) {
    uint32_t rl = (uint32_t)r, rh = (uint32_t)(r >> 32), t;
    asm volatile("v_cndmask_b32_e64 %2, 0, -1, %3\n\t"          // e in the flagged lanes
                 "v_sub_co_u32_e32 %0, vcc, %0, %2\n\t"         // r - e
                 "v_subbrev_co_u32_e32 %1, vcc, 0, %1, vcc\n\t"
                 "v_cndmask_b32_e64 %2, 0, -1, vcc\n\t"          // borrowed (r < e): -2^64 = -e
                 "v_sub_co_u32_e32 %0, vcc, %0, %2\n\t"
                 "v_subbrev_co_u32_e32 %1, vcc, 0, %1, vcc"       // cannot borrow again: the wrapped r - e is above 2^64 - e
                 : "+v"(rl), "+v"(rh), "=&v"(t)
                 : "s"(f)
                 : "vcc");
    return ((uint64_t)rh << 32) | rl;
}

// One radix-2 butterfly, s = a + c and d = a - c (NEG: c - a), as ONE block with both rare paths deferred: the nine instructions of
// add_f and sub_f without the wait state the compiler puts between two asm statements.
template <bool NEG>
__device__ __forceinline__ void bfly_f(uint64_t a, uint64_t c, uint64_t &s, uint64_t &d, rare_mask &fa, rare_mask &fs) {
    const uint32_t al = (uint32_t)a, ah = (uint32_t)(a >> 32), cl = (uint32_t)c, ch = (uint32_t)(c >> 32);
    uint32_t dl, dh, t;
    uint64_t r;
    asm("v_add_co_u32_e32 v32, vcc, %6, %8\n\t"
        "v_addc_co_u32_e32 v33, vcc, %7, %9, vcc\n\t"
        "v_cndmask_b32_e64 v42, 0, -1, vcc\n\t"
        "v_mad_u64_u32 %0, %4, v42, 1, v[32:33]\n\t"             // s, second wrap -> fa
        "v_sub_co_u32_e32 %1, vcc, %10, %12\n\t"
        "v_subb_co_u32_e32 %2, vcc, %11, %13, vcc\n\t"
        "v_cndmask_b32_e64 %3, 0, -1, vcc\n\t"
        "v_sub_co_u32_e32 %1, vcc, %1, %3\n\t"
        "v_subbrev_co_u32_e64 %2, %5, 0, %2, vcc"                   // d, second borrow -> fs
        : "=&v"(r), "=&v"(dl), "=&v"(dh), "=&v"(t), "=&s"(fa), "=&s"(fs)
        : "v"(al), "v"(ah), "v"(cl), "v"(ch), "v"(NEG ? cl : al), "v"(NEG ? ch : ah), "v"(NEG ? al : cl), "v"(NEG ? ah : ch)
        : "vcc", "v32", "v33", "v42");
    s = r;
    d = ((uint64_t)dh << 32) | dl;
}

// x * 2^K with the rare borrow deferred (0 <= K < 96). The forms that have no rare path (K a multiple of 32, K < 32) return mask 0 — a
// compile-time constant that the caller's OR folds away.
template <int K>
__device__ __forceinline__ uint64_t mul_pow2_f(uint64_t x, rare_mask &f);
template <int K>
__device__ __forceinline__ uint64_t mul_pow2_fix(uint64_t r, rare_mask f) {
    constexpr int Q = K / 32, S = K % 32;
    if constexpr (S != 0 && Q >= 1) return sub_fix(r, f);  // the deferred correction is "- e once more", as sub's
    return r;
}

// x in the flagged lanes, 0 in the others
__device__ __forceinline__ uint64_t masked(uint64_t x, rare_mask f) {
    uint32_t xl = (uint32_t)x, xh = (uint32_t)(x >> 32), rl, rh;
    asm volatile("v_cndmask_b32_e64 %0, 0, %2, %4\n\t"
                 "v_cndmask_b32_e64 %1, 0, %3, %4"
                 : "=&v"(rl), "=&v"(rh)
                 : "v"(xl), "v"(xh), "s"(f));
    return ((uint64_t)rh << 32) | rl;
}
// the compile-time constant C in the flagged lanes, 0 in the others. The constant is materialised HERE, by the asm itself: a
// constant the compiler sees is hoisted out of the (cold) block that uses it and out of the loop around it, and then occupies
// vector registers through the hot path.
template <uint64_t C>
__device__ __forceinline__ uint64_t masked_const(rare_mask f) {
    uint32_t rl, rh;
    asm volatile("v_mov_b32_e32 %0, %2\n\t"
                 "v_mov_b32_e32 %1, %3\n\t"
                 "v_cndmask_b32_e64 %0, 0, %0, %4\n\t"
                 "v_cndmask_b32_e64 %1, 0, %1, %4"
                 : "=&v"(rl), "=&v"(rh)
                 : "i"((uint32_t)C), "i"((uint32_t)(C >> 32)), "s"(f));
    return ((uint64_t)rh << 32) | rl;
}
// (2^32 - 1) 2^k mod p, canonical, for 0 <= k < 96
constexpr uint64_t eps_times_pow2(int k) {
    uint64_t x = 0xFFFFFFFFull;
    for (int i = 0; i < k; i++) {
        const bool top = x >> 63;
        x <<= 1;                       // 2 x mod 2^64; the lost 2^64 = 2^32 - 1 (mod p)
        if (top) x += 0xFFFFFFFFull;   // x < 2^64 - 2^33 + 2 before, so no wrap (x was canonical: 2 x - 2^64 < p - 2^32)
        if (x >= 0xFFFFFFFF00000001ull) x -= 0xFFFFFFFF00000001ull;
    }
    return x;
}

// did any lane of the wave flag any operation of the group? (uniform: the masks are scalar registers)
// A MACRO, not a function: the hint must sit in the `if` itself. (Returned from an inline function it is dropped before inlining, the
// compiler lays the correction block out as the fall-through, and the hot path pays a TAKEN branch over it at every group.)
#define GL_RARE_ANY(m) __builtin_expect((m) != 0, 0)

// acc + x*y (goldilocks_field.rs:119-123): the multiplication followed by the addition, 16 VALU; the fused
// arrangement this replaced (the addend riding on the first multiply-adds) took 22.
__device__ __forceinline__ uint64_t mac(uint64_t acc, uint64_t x, uint64_t y) { return add(acc, mul(x, y)); }

// Two radix-2 butterflies at once on arbitrary representatives:
//   s0 = a0 + c0, d0 = x0 - y0, s1 = a1 + c1, d1 = x1 - y1     (x,y = a,c or c,a when NEG)
// The first version interleaved the four carry chains with the full double wrap correction in one block of 32
// carry-flag instructions (~148 cycles per wavefront by tools/ubench_issue.hip's per-instruction costs); built
// from add / sub above — second correction behind a never-taken branch — it is 18 VALU and ~106 cycles.
template <bool NEG0, bool NEG1>
__device__ __forceinline__ void bfly2(uint64_t a0, uint64_t c0, uint64_t a1, uint64_t c1, uint64_t &s0, uint64_t &d0,
                                      uint64_t &s1, uint64_t &d1) {
    s0 = add(a0, c0);
    d0 = NEG0 ? sub(c0, a0) : sub(a0, c0);
    s1 = add(a1, c1);
    d1 = NEG1 ? sub(c1, a1) : sub(a1, c1);
}

// canonical-output product (inputs may be any u64)
__device__ __forceinline__ uint64_t mul_c(uint64_t a, uint64_t b) {
    uint64_t lo, hi;
    mul_wide(a, b, lo, hi);
    return reduce128_c(lo, hi);
}

// x * 2^k mod p for a compile-time 0 <= k < 192, multiply-free; x any u64, result any u64.
// 2^96 = -1 (mod p), so k >= 96 is the negated shift by k-96 (callers that can absorb the sign —
// the NTT butterflies — never take that branch). For k = 32q + s < 96 the 64-bit x shifted left
// by s is a 96-bit number (w2:w1:w0), and with 2^64 = 2^32-1 =: e, 2^96 = -1, 2^128 = -2^32:
//   q = 0:  (w1:w0) + w2*e
//   q = 1:  (w0:0)  + w1*e - w2
//   q = 2:   w0*e   - (w2:w1)
// Every line needs at most one wrap correction per add/sub (bounds in the comments below).
template <int K>
__device__ __forceinline__ uint64_t mul_pow2(uint64_t x) {
    static_assert(K >= 0 && K < 192, "shift out of range");
    if constexpr (K == 0) {
        return x;
    } else if constexpr (K >= 96) {
        return neg(mul_pow2<K - 96>(x));
    } else {
        constexpr int Q = K / 32, S = K % 32;
        const uint32_t xl = (uint32_t)x, xh = (uint32_t)(x >> 32);
        if constexpr (S != 0) {
            // The three lines above on the carry flags (6 / 9 / 6 instructions plus rare paths; the portable form below
            // compiles to 64-bit compares and adds that cost about twice as many issue slots). These are
            // the shift twiddles of the radix-16 butterflies: w_16 = 2^12, so K is a multiple of 12.
            // The wrap correction after a multiply-add is itself a multiply-add (t*1 + r: 6 cycles against 10 for
            // v_add_co + v_addc_co); the borrow corrections trigger with probability <= 2^(S-32) per lane
            // (Q = 2) or ~2^(S-64) (Q = 1) and sit behind a wave-uniform branch, as in add / sub / mul.
            uint32_t rl, rh, t;
            if constexpr (Q == 0) {
                uint64_t r;
                asm("v_lshlrev_b32_e32 v32, %[s], %[xl]\n\t"                  // w0
                    "v_alignbit_b32 v33, %[xh], %[xl], %[r]\n\t"              // w1
                    "v_lshrrev_b32_e32 %[t], %[r], %[xh]\n\t"                  // w2
                    "v_mad_u64_u32 v[32:33], vcc, %[t], -1, v[32:33]\n\t"  // (w1:w0) + w2*e
                    "v_cndmask_b32_e64 %[t], 0, -1, vcc\n\t"
                    "v_mad_u64_u32 %[out], vcc, %[t], 1, v[32:33]"             // wrapped: + e (cannot wrap again)
                    : [out] "=&v"(r), [t] "=&v"(t)
                    : [xl] "v"(xl), [xh] "v"(xh), [s] "n"(S), [r] "n"(32 - S)
                    : "vcc", "v32", "v33");
                return r;
            } else if constexpr (Q == 1) {
                asm("v_mov_b32_e32 v32, 0\n\t"
                    "v_lshlrev_b32_e32 v33, %[s], %[xl]\n\t"                  // (w0:0)
                    "v_alignbit_b32 %[t], %[xh], %[xl], %[r]\n\t"              // w1
                    "v_mad_u64_u32 v[32:33], vcc, %[t], -1, v[32:33]\n\t"  // + w1*e
                    "v_cndmask_b32_e64 %[t], 0, -1, vcc\n\t"
                    "v_mad_u64_u32 v[32:33], vcc, %[t], 1, v[32:33]\n\t"   // wrapped: + e
                    "v_lshrrev_b32_e32 %[t], %[r], %[xh]\n\t"                  // w2
                    "v_sub_co_u32_e32 %[rl], vcc, v32, %[t]\n\t"              // - w2
                    "v_subbrev_co_u32_e32 %[rh], vcc, 0, v33, vcc\n\t"
                    "s_cbranch_vccz 1f\n\t"
                    "v_cndmask_b32_e64 %[t], 0, -1, vcc\n\t"                   // rare: borrow => - e
                    "v_sub_co_u32_e32 %[rl], vcc, %[rl], %[t]\n\t"
                    "v_subbrev_co_u32_e32 %[rh], vcc, 0, %[rh], vcc\n\t"
                    "1:"
                    : [rl] "=&v"(rl), [rh] "=&v"(rh), [t] "=&v"(t)
                    : [xl] "v"(xl), [xh] "v"(xh), [s] "n"(S), [r] "n"(32 - S)
                    : "vcc", "v32", "v33");
            } else {
                uint32_t u;
                asm("v_lshlrev_b32_e32 %[t], %[s], %[xl]\n\t"                  // w0
                    "v_mad_u64_u32 v[32:33], vcc, %[t], -1, 0\n\t"           // w0*e
                    "v_alignbit_b32 %[t], %[xh], %[xl], %[r]\n\t"              // w1
                    "v_lshrrev_b32_e32 %[u], %[r], %[xh]\n\t"                  // w2
                    "v_sub_co_u32_e32 %[rl], vcc, v32, %[t]\n\t"              // - (w2:w1)
                    "v_subb_co_u32_e32 %[rh], vcc, v33, %[u], vcc\n\t"
                    "s_cbranch_vccz 1f\n\t"
                    "v_cndmask_b32_e64 %[t], 0, -1, vcc\n\t"                   // rare: borrow => - e
                    "v_sub_co_u32_e32 %[rl], vcc, %[rl], %[t]\n\t"
                    "v_subbrev_co_u32_e32 %[rh], vcc, 0, %[rh], vcc\n\t"
                    "1:"
                    : [rl] "=&v"(rl), [rh] "=&v"(rh), [t] "=&v"(t), [u] "=&v"(u)
                    : [xl] "v"(xl), [xh] "v"(xh), [s] "n"(S), [r] "n"(32 - S)
                    : "vcc", "v32", "v33");
            }
            return pack64(rl, rh);
        }
        uint32_t w0, w1, w2;
        if constexpr (S == 0) {
            w0 = xl; w1 = xh; w2 = 0;
        } else {
            w0 = xl << S;
            w1 = __builtin_amdgcn_alignbit(xh, xl, 32 - S);
            w2 = xh >> (32 - S);
        }
        auto times_eps = [](uint32_t w) { return ((uint64_t)w << 32) - w; };  // w * (2^32 - 1) < 2^64
        if constexpr (Q == 0) {
            uint64_t lo = ((uint64_t)w1 << 32) | w0, t = times_eps(w2);
            uint64_t r = lo + t;
            return r + ((r < t) ? EPS : 0);  // wrapped r < t <= (2^32-1)^2, so r + e cannot wrap again
        } else if constexpr (Q == 1) {
            uint64_t a = (uint64_t)w0 << 32, t = times_eps(w1);
            uint64_t r = a + t;
            r += (r < t) ? EPS : 0;
            uint64_t r2 = r - w2;
            return r2 - ((r < (uint64_t)w2) ? EPS : 0);  // borrow => r2 >= 2^64 - 2^32 > e
        } else {
            uint64_t t = times_eps(w0), u = ((uint64_t)w2 << 32) | w1;
            uint64_t r = t - u;
            return r - ((t < u) ? EPS : 0);  // borrow => r >= 2^64 - u > 2^63 > e (u < 2^63)
        }
    }
}

template <int K>
__device__ __forceinline__ uint64_t mul_pow2_f(uint64_t x, rare_mask &f) {
    static_assert(K >= 0 && K < 96, "shift out of range");
    constexpr int Q = K / 32, S = K % 32;
    if constexpr (S == 0 || Q == 0) {
        f = 0;
        return mul_pow2<K>(x);
    } else {
        const uint32_t xl = (uint32_t)x, xh = (uint32_t)(x >> 32);
        uint32_t rl, rh, t;
        if constexpr (Q == 1) {
            asm("v_mov_b32_e32 v32, 0\n\t"
                "v_lshlrev_b32_e32 v33, %[s], %[xl]\n\t"                  // (w0:0)
                "v_alignbit_b32 %[t], %[xh], %[xl], %[r]\n\t"              // w1
                "v_mad_u64_u32 v[32:33], vcc, %[t], -1, v[32:33]\n\t"  // + w1*e
                "v_cndmask_b32_e64 %[t], 0, -1, vcc\n\t"
                "v_mad_u64_u32 v[32:33], vcc, %[t], 1, v[32:33]\n\t"   // wrapped: + e
                "v_lshrrev_b32_e32 %[t], %[r], %[xh]\n\t"                  // w2
                "v_sub_co_u32_e32 %[rl], vcc, v32, %[t]\n\t"              // - w2
                "v_subbrev_co_u32_e64 %[rh], %[f], 0, v33, vcc"              // borrow -> the mask
                : [rl] "=&v"(rl), [rh] "=&v"(rh), [t] "=&v"(t), [f] "=&s"(f)
                : [xl] "v"(xl), [xh] "v"(xh), [s] "n"(S), [r] "n"(32 - S)
                : "vcc", "v32", "v33");
        } else {
            uint32_t u;
            asm("v_lshlrev_b32_e32 %[t], %[s], %[xl]\n\t"                  // w0
                "v_mad_u64_u32 v[32:33], vcc, %[t], -1, 0\n\t"           // w0*e
                "v_alignbit_b32 %[t], %[xh], %[xl], %[r]\n\t"              // w1
                "v_lshrrev_b32_e32 %[u], %[r], %[xh]\n\t"                  // w2
                "v_sub_co_u32_e32 %[rl], vcc, v32, %[t]\n\t"              // - (w2:w1)
                "v_subb_co_u32_e64 %[rh], %[f], v33, %[u], vcc"              // borrow -> the mask
                : [rl] "=&v"(rl), [rh] "=&v"(rh), [t] "=&v"(t), [u] "=&v"(u), [f] "=&s"(f)
                : [xl] "v"(xl), [xh] "v"(xh), [s] "n"(S), [r] "n"(32 - S)
                : "vcc", "v32", "v33");
        }
        return pack64(rl, rh);
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

// al + ah*2^32 (mod p) for any al and ah < 2^63 — the two 64-bit column sums of an MDS row (poseidon.h) and of
// the gate programs' ACC accumulators. (h = ah.hi + carry <= 2^31, so after the one possible wrap of
// l + h*(2^32 - 1) the value is below h*2^32 <= 2^63 and adding 2^32 - 1 cannot wrap again.) The compiler's version of "fold into (lo64, hi32), then reduce96" is ~18 issue
// slots of double-pumped 64-bit compares and adds; with the carry flags it is 7 single instructions:
//   l = al + (ah << 32)  ->  l.hi = al.hi + ah.lo (carry c), h = ah.hi + c
//   r = l + h*(2^32 - 1) as one v_mad_u64_u32, its carry-out adds 2^32 - 1 once more (cannot carry again).
__device__ __forceinline__ uint64_t fold96(uint64_t al, uint64_t ah) {
    uint32_t all = (uint32_t)al, alh = (uint32_t)(al >> 32), ahl = (uint32_t)ah, ahh = (uint32_t)(ah >> 32);
    uint64_t r;
    asm("v_mov_b32_e32 v32, %1\n\t"                                     // l = al + (ah << 32): (al.lo, al.hi + ah.lo), carry
        "v_add_co_u32_e32 v33, vcc, %2, %3\n\t"
        "v_addc_co_u32_e32 v34, vcc, 0, %4, vcc\n\t"                   // h = ah.hi + carry
        "v_mad_u64_u32 v[32:33], vcc, v34, -1, v[32:33]\n\t"      // l + h*(2^32-1)
        "v_cndmask_b32_e64 v34, 0, -1, vcc\n\t"
        "v_mad_u64_u32 %0, vcc, v34, 1, v[32:33]"                     // the wrap correction as a multiply-add
        : "=&v"(r)
        : "v"(all), "v"(alh), "v"(ahl), "v"(ahh)
        : "vcc", "v32", "v33", "v34");
    return r;
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

// Carry-free dot products for x times COMPILE-TIME constants (Poseidon's partial rounds). x is split once into limbs of
// 21 / 21 / 22 bits, x = x0 + x1*2^21 + x2*2^42, and every constant c comes with c*2^21 and c*2^42 (mod p) from a table:
//     x*c = x0*c + x1*(c 2^21) + x2*(c 2^42)   (mod p).
// With the three constants split into 32-bit halves every partial product is below 2^54, so a dot product of up to ~500 terms
// is two plain 64-bit sums (low halves, high halves): six v_mad_u64_u32 per term and NO carry instruction (dot_term needs
// four multiply-adds plus four carry counters), two accumulator registers pairs instead of 4.5, and the reduction at the end
// is fold96 (7 instructions) instead of dot_finish (20). The vector ALU of the permutation kernel is issue-bound at four
// cycles per instruction whatever the instruction (rocprofv3: SQ_INSTS_VALU x 4 = kernel cycles), so instructions saved are
// time saved.
struct Limbs3 {
    uint32_t x0, x1, x2;
};
__device__ __forceinline__ Limbs3 split21(uint64_t x) {
    Limbs3 r;
    r.x0 = (uint32_t)x & 0x1FFFFFu;
    r.x1 = (uint32_t)(x >> 21) & 0x1FFFFFu;
    r.x2 = (uint32_t)(x >> 42);
    return r;
}
struct DotAcc2 {
    uint64_t lo = 0, hi = 0;  // value = lo + hi * 2^32
};
__device__ __forceinline__ void dot_term3(DotAcc2 &d, const Limbs3 &x, uint64_t c0, uint64_t c1, uint64_t c2) {
    d.lo += (uint64_t)x.x0 * (uint32_t)c0;
    d.hi += (uint64_t)x.x0 * (uint32_t)(c0 >> 32);
    d.lo += (uint64_t)x.x1 * (uint32_t)c1;
    d.hi += (uint64_t)x.x1 * (uint32_t)(c1 >> 32);
    d.lo += (uint64_t)x.x2 * (uint32_t)c2;
    d.hi += (uint64_t)x.x2 * (uint32_t)(c2 >> 32);
}
// a term with a small constant (< 2^32) needs no limbs: x * c = xl*c + (xh*c) * 2^32, both products below 2^64 / terms
__device__ __forceinline__ void dot_term_small(DotAcc2 &d, uint64_t x, uint32_t c) {
    d.lo += (uint64_t)(uint32_t)x * c;
    d.hi += (uint64_t)(uint32_t)(x >> 32) * c;
}

// value = (A0 + k0*2^64) + (A1 + k1*2^64)*2^32 + (A2 + k2*2^64)*2^64   (mod p): the portable statement of
// what dot_finish (carry-flag version, below) computes.
__device__ __forceinline__ uint64_t dot_finish_generic(const DotAcc &d) {
    uint64_t lo = d.a0 + (d.a1 << 32);
    uint64_t c0 = lo < d.a0;
    uint64_t h1 = (d.a1 >> 32) + c0 + d.k0;  // < 2^33, no wrap
    uint64_t hi = h1 + d.a2;
    uint64_t top = (uint64_t)(hi < h1) + d.k2;  // units of 2^128
    uint64_t k1s = (uint64_t)d.k1 << 32;  // k1 * 2^96 = (k1 << 32) * 2^64
    uint64_t hi2 = hi + k1s;
    top += hi2 < k1s;
    // 2^128 = -2^32 (mod p); top < 2^7 so top << 32 is canonical
    return sub(reduce128(lo, hi2), top << 32);
}

// The same reduction on the carry flags, 20 instructions instead of ~30 issue slots of double-pumped
// 64-bit compares. In 32-bit words the accumulated value is
//   w0 + w1*2^32 + w2*2^64 + w3*2^96 + w4*2^128,   w0 = a0.lo, w1 = a0.hi + a1.lo, w2 = a1.hi + a2.lo + k0,
//   w3 = a2.hi + k1, w4 = k2 (+ carries; w4 stays tiny: it counts terms),
// and 2^64 = 2^32-1, 2^96 = -1, 2^128 = -2^32 (mod p) give  (w0,w1) - w3 + w2*(2^32-1) - w4*2^32.
// Every wrap is corrected once and cannot wrap again: after a borrow the value is >= 2^64 - 2^42, after
// the multiply-add's carry it is < w2*(2^32-1).
__device__ __forceinline__ uint64_t dot_finish(const DotAcc &d) {
    uint32_t a0l = (uint32_t)d.a0, a0h = (uint32_t)(d.a0 >> 32), a1l = (uint32_t)d.a1, a1h = (uint32_t)(d.a1 >> 32);
    uint32_t a2l = (uint32_t)d.a2, a2h = (uint32_t)(d.a2 >> 32);
    uint32_t rl, rh, w1, w2, w3, w4, t;
    asm("v_add_co_u32_e32 %[w1], vcc, %[a0h], %[a1l]\n\t"
        "v_addc_co_u32_e32 %[w2], vcc, %[a1h], %[a2l], vcc\n\t"
        "v_addc_co_u32_e32 %[w3], vcc, %[a2h], %[k1], vcc\n\t"
        "v_addc_co_u32_e32 %[w4], vcc, 0, %[k2], vcc\n\t"
        "v_add_co_u32_e32 %[w2], vcc, %[w2], %[k0]\n\t"
        "v_addc_co_u32_e32 %[w3], vcc, 0, %[w3], vcc\n\t"
        "v_addc_co_u32_e32 %[w4], vcc, 0, %[w4], vcc\n\t"
        "v_sub_co_u32_e32 v32, vcc, %[a0l], %[w3]\n\t"               // (w0, w1) - w3
        "v_subbrev_co_u32_e32 v33, vcc, 0, %[w1], vcc\n\t"
        "v_cndmask_b32_e64 %[t], 0, -1, vcc\n\t"
        "v_sub_co_u32_e32 v32, vcc, v32, %[t]\n\t"
        "v_subbrev_co_u32_e32 v33, vcc, 0, v33, vcc\n\t"
        "v_mad_u64_u32 v[32:33], vcc, %[w2], -1, v[32:33]\n\t"    // + w2 * (2^32 - 1)
        "v_cndmask_b32_e64 %[t], 0, -1, vcc\n\t"
        "v_add_co_u32_e32 v32, vcc, v32, %[t]\n\t"
        "v_addc_co_u32_e32 v33, vcc, 0, v33, vcc\n\t"
        "v_sub_co_u32_e32 v33, vcc, v33, %[w4]\n\t"                 // - w4 * 2^32
        "v_cndmask_b32_e64 %[t], 0, -1, vcc\n\t"
        "v_sub_co_u32_e32 %[rl], vcc, v32, %[t]\n\t"
        "v_subbrev_co_u32_e32 %[rh], vcc, 0, v33, vcc"
        : [rl] "=&v"(rl), [rh] "=&v"(rh), [w1] "=&v"(w1), [w2] "=&v"(w2), [w3] "=&v"(w3), [w4] "=&v"(w4), [t] "=&v"(t)
        : [a0l] "v"(a0l), [a0h] "v"(a0h), [a1l] "v"(a1l), [a1h] "v"(a1h), [a2l] "v"(a2l), [a2h] "v"(a2h), [k0] "v"(d.k0),
          [k1] "v"(d.k1), [k2] "v"(d.k2)
        : "vcc", "v32", "v33");
    return pack64(rl, rh);
}


}  // namespace gl

#ifndef GL_JIT
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
#endif  // GL_JIT
