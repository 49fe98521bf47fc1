typedef unsigned long long uint64_t; typedef unsigned int uint32_t; typedef unsigned short uint16_t; typedef long long int64_t; typedef int int32_t;
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
// corrections built from double-pumped 64-bit ops. Here (18 VALU):
//   product:  T = al*bl ; U = al*bh + (T>>32) ; V = ah*bl + (U.lo,0) ; W = ah*bh + (U>>32) + V.hi
//             -> lo = (T.lo, V.lo), hi = W           (no intermediate can overflow 64 bits)
//   reduce :  t0 = lo - hh (borrow => -= 2^32-1) ; r = t0 + hl*(2^32-1) as ONE v_mad_u64_u32
//             whose carry-out drives the last correction (goldilocks_field.rs:345-358).
// `s_nop 1` = the two wait states between a VALU instruction that writes VCC/an SGPR and the VALU
// instruction that consumes it as carry-in or select mask.
__device__ __forceinline__ uint64_t mul(uint64_t a, uint64_t b) {
    uint32_t al = (uint32_t)a, ah = (uint32_t)(a >> 32), bl = (uint32_t)b, bh = (uint32_t)(b >> 32);
    uint32_t rl, rh;
    // LLVM's AMDGPU inline asm has no sub-register operand modifier, so the 64-bit temporaries
    // whose halves are needed live in fixed registers v[116:126] (declared clobbered).
    asm("v_mad_u64_u32 v[116:117], vcc, %2, %4, 0\n\t"          // T = al*bl
        "v_mov_b32_e32 v125, 0\n\t"
        "v_mov_b32_e32 v124, v117\n\t"                          // X = (T.hi, 0)
        "v_mad_u64_u32 v[118:119], vcc, %2, %5, v[124:125]\n\t" // U = al*bh + T.hi
        "v_mov_b32_e32 v124, v119\n\t"                          // X = (U.hi, 0)
        "v_mad_u64_u32 v[122:123], vcc, %3, %5, v[124:125]\n\t" // W = ah*bh + U.hi
        "v_mov_b32_e32 v124, v118\n\t"                          // X = (U.lo, 0)
        "v_mad_u64_u32 v[120:121], vcc, %3, %4, v[124:125]\n\t" // V = ah*bl + U.lo = (lo.hi, carry)
        "v_add_co_u32_e32 v122, vcc, v122, v121\n\t"            // W += V.hi   -> hi = (hl, hh) = (v122, v123)
        "s_nop 1\n\t"
        "v_addc_co_u32_e32 v123, vcc, 0, v123, vcc\n\t"
        "v_sub_co_u32_e32 v116, vcc, v116, v123\n\t"            // t0 = lo - hh, lo = (v116, v120)
        "s_nop 1\n\t"
        "v_subbrev_co_u32_e32 v117, vcc, 0, v120, vcc\n\t"
        "s_nop 1\n\t"
        "v_cndmask_b32_e64 v126, 0, -1, vcc\n\t"                // borrow: t0 -= 2^32-1
        "v_sub_co_u32_e32 v116, vcc, v116, v126\n\t"
        "s_nop 1\n\t"
        "v_subbrev_co_u32_e32 v117, vcc, 0, v117, vcc\n\t"
        "v_mad_u64_u32 v[116:117], vcc, v122, -1, v[116:117]\n\t"  // r = t0 + hl*(2^32-1), carry -> vcc
        "s_nop 1\n\t"
        "v_cndmask_b32_e64 v126, 0, -1, vcc\n\t"                // carry: r += 2^32-1 (cannot carry again)
        "v_add_co_u32_e32 %0, vcc, v116, v126\n\t"
        "s_nop 1\n\t"
        "v_addc_co_u32_e32 %1, vcc, 0, v117, vcc"
        : "=&v"(rl), "=&v"(rh)
        : "v"(al), "v"(ah), "v"(bl), "v"(bh)
        : "vcc", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126");
    return pack64(rl, rh);
}

__device__ __forceinline__ uint64_t sqr(uint64_t a) { return mul(a, a); }

// acc + x*y (goldilocks_field.rs:119-123); u64 + u64*u64 cannot overflow 128 bits.
// Same schedule as mul(): the addend's low word rides on the first v_mad_u64_u32 and its high
// word joins T.hi before the second one (al*bh + T.hi + c.hi <= 2^64 - 1).
__device__ __forceinline__ uint64_t mac(uint64_t acc, uint64_t x, uint64_t y) {
    uint32_t al = (uint32_t)x, ah = (uint32_t)(x >> 32), bl = (uint32_t)y, bh = (uint32_t)(y >> 32);
    uint32_t cl = (uint32_t)acc, ch = (uint32_t)(acc >> 32);
    uint32_t rl, rh;
    asm("v_mov_b32_e32 v125, 0\n\t"
        "v_mov_b32_e32 v124, %6\n\t"                            // X = (c.lo, 0)
        "v_mad_u64_u32 v[116:117], vcc, %2, %4, v[124:125]\n\t" // T = al*bl + c.lo
        "v_add_co_u32_e32 v124, vcc, v117, %7\n\t"              // X = T.hi + c.hi (33 bits)
        "s_nop 1\n\t"
        "v_addc_co_u32_e32 v125, vcc, 0, v125, vcc\n\t"
        "v_mad_u64_u32 v[118:119], vcc, %2, %5, v[124:125]\n\t" // U = al*bh + X
        "v_mov_b32_e32 v125, 0\n\t"
        "v_mov_b32_e32 v124, v119\n\t"                          // X = (U.hi, 0)
        "v_mad_u64_u32 v[122:123], vcc, %3, %5, v[124:125]\n\t" // W = ah*bh + U.hi
        "v_mov_b32_e32 v124, v118\n\t"                          // X = (U.lo, 0)
        "v_mad_u64_u32 v[120:121], vcc, %3, %4, v[124:125]\n\t" // V = ah*bl + U.lo
        "v_add_co_u32_e32 v122, vcc, v122, v121\n\t"            // W += V.hi
        "s_nop 1\n\t"
        "v_addc_co_u32_e32 v123, vcc, 0, v123, vcc\n\t"
        "v_sub_co_u32_e32 v116, vcc, v116, v123\n\t"            // t0 = lo - hh
        "s_nop 1\n\t"
        "v_subbrev_co_u32_e32 v117, vcc, 0, v120, vcc\n\t"
        "s_nop 1\n\t"
        "v_cndmask_b32_e64 v126, 0, -1, vcc\n\t"
        "v_sub_co_u32_e32 v116, vcc, v116, v126\n\t"
        "s_nop 1\n\t"
        "v_subbrev_co_u32_e32 v117, vcc, 0, v117, vcc\n\t"
        "v_mad_u64_u32 v[116:117], vcc, v122, -1, v[116:117]\n\t"
        "s_nop 1\n\t"
        "v_cndmask_b32_e64 v126, 0, -1, vcc\n\t"
        "v_add_co_u32_e32 %0, vcc, v116, v126\n\t"
        "s_nop 1\n\t"
        "v_addc_co_u32_e32 %1, vcc, 0, v117, vcc"
        : "=&v"(rl), "=&v"(rh)
        : "v"(al), "v"(ah), "v"(bl), "v"(bh), "v"(cl), "v"(ch)
        : "vcc", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126");
    return pack64(rl, rh);
}

// Two radix-2 butterflies at once on arbitrary representatives:
//   s0 = a0 + c0, d0 = x0 - y0, s1 = a1 + c1, d1 = x1 - y1     (x,y = a,c or c,a when NEG)
// with the reference's double wrap correction (goldilocks_field.rs:197-256) done on the carry
// flags. The four carry chains are interleaved instruction by instruction, so every
// VALU-writes-SGPR -> VALU-reads-it dependency has three independent instructions in between
// (gfx950 needs two wait states) and no s_nop is spent; 32 VALU instructions for what the
// compiler's compare/select lowering does in ~48 issue slots.
#define GL_BFLY2_ASM(X0L, X0H, X1L, X1H)                                                          \
    /* raw 64-bit sums / differences */                                                            \
    "v_add_co_u32_e64 %0, %12, %16, %18\n\t"                                                       \
    "v_sub_co_u32_e64 %2, %13, " X0L "\n\t"                                                        \
    "v_add_co_u32_e64 %4, %14, %20, %22\n\t"                                                       \
    "v_sub_co_u32_e64 %6, %15, " X1L "\n\t"                                                        \
    "v_addc_co_u32_e64 %1, %12, %17, %19, %12\n\t"                                                 \
    "v_subb_co_u32_e64 %3, %13, " X0H ", %13\n\t"                                                  \
    "v_addc_co_u32_e64 %5, %14, %21, %23, %14\n\t"                                                 \
    "v_subb_co_u32_e64 %7, %15, " X1H ", %15\n\t" /* first correction: +/- (2^32 - 1) on carry / borrow */ \
    "v_cndmask_b32_e64 %8, 0, -1, %12\n\t"                                                         \
    "v_cndmask_b32_e64 %9, 0, -1, %13\n\t"                                                         \
    "v_cndmask_b32_e64 %10, 0, -1, %14\n\t"                                                        \
    "v_cndmask_b32_e64 %11, 0, -1, %15\n\t"                                                        \
    "v_add_co_u32_e64 %0, %12, %0, %8\n\t"                                                         \
    "v_sub_co_u32_e64 %2, %13, %2, %9\n\t"                                                         \
    "v_add_co_u32_e64 %4, %14, %4, %10\n\t"                                                        \
    "v_sub_co_u32_e64 %6, %15, %6, %11\n\t"                                                        \
    "v_addc_co_u32_e64 %1, %12, 0, %1, %12\n\t"                                                    \
    "v_subb_co_u32_e64 %3, %13, %3, 0, %13\n\t"                                                    \
    "v_addc_co_u32_e64 %5, %14, 0, %5, %14\n\t"                                                    \
    "v_subb_co_u32_e64 %7, %15, %7, 0, %15\n\t" /* second (rare) correction */                     \
    "v_cndmask_b32_e64 %8, 0, -1, %12\n\t"                                                         \
    "v_cndmask_b32_e64 %9, 0, -1, %13\n\t"                                                         \
    "v_cndmask_b32_e64 %10, 0, -1, %14\n\t"                                                        \
    "v_cndmask_b32_e64 %11, 0, -1, %15\n\t"                                                        \
    "v_add_co_u32_e64 %0, %12, %0, %8\n\t"                                                         \
    "v_sub_co_u32_e64 %2, %13, %2, %9\n\t"                                                         \
    "v_add_co_u32_e64 %4, %14, %4, %10\n\t"                                                        \
    "v_sub_co_u32_e64 %6, %15, %6, %11\n\t"                                                        \
    "v_addc_co_u32_e64 %1, %12, 0, %1, %12\n\t"                                                    \
    "v_subb_co_u32_e64 %3, %13, %3, 0, %13\n\t"                                                    \
    "v_addc_co_u32_e64 %5, %14, 0, %5, %14\n\t"                                                    \
    "v_subb_co_u32_e64 %7, %15, %7, 0, %15"

template <bool NEG0, bool NEG1>
__device__ __forceinline__ void bfly2(uint64_t a0, uint64_t c0, uint64_t a1, uint64_t c1, uint64_t &s0, uint64_t &d0,
                                      uint64_t &s1, uint64_t &d1) {
    uint32_t a0l = (uint32_t)a0, a0h = (uint32_t)(a0 >> 32), c0l = (uint32_t)c0, c0h = (uint32_t)(c0 >> 32);
    uint32_t a1l = (uint32_t)a1, a1h = (uint32_t)(a1 >> 32), c1l = (uint32_t)c1, c1h = (uint32_t)(c1 >> 32);
    uint32_t s0l, s0h, d0l, d0h, s1l, s1h, d1l, d1h, e0, e1, e2, e3;
    uint64_t k0, k1, k2, k3;
#define GL_BFLY2_OPERANDS                                                                                        \
    : "=&v"(s0l), "=&v"(s0h), "=&v"(d0l), "=&v"(d0h), "=&v"(s1l), "=&v"(s1h), "=&v"(d1l), "=&v"(d1h), /* 0-7 */  \
      "=&v"(e0), "=&v"(e1), "=&v"(e2), "=&v"(e3),                                                     /* 8-11 */ \
      "=&s"(k0), "=&s"(k1), "=&s"(k2), "=&s"(k3)                                                      /* 12-15 */ \
    : "v"(a0l), "v"(a0h), "v"(c0l), "v"(c0h), "v"(a1l), "v"(a1h), "v"(c1l), "v"(c1h)                  /* 16-23 */
    // difference operands: a - c, or c - a when the twiddle's sign is absorbed (NEG)
    if constexpr (!NEG0 && !NEG1)
        asm(GL_BFLY2_ASM("%16, %18", "%17, %19", "%20, %22", "%21, %23") GL_BFLY2_OPERANDS);
    else if constexpr (NEG0 && !NEG1)
        asm(GL_BFLY2_ASM("%18, %16", "%19, %17", "%20, %22", "%21, %23") GL_BFLY2_OPERANDS);
    else if constexpr (!NEG0 && NEG1)
        asm(GL_BFLY2_ASM("%16, %18", "%17, %19", "%22, %20", "%23, %21") GL_BFLY2_OPERANDS);
    else
        asm(GL_BFLY2_ASM("%18, %16", "%19, %17", "%22, %20", "%23, %21") GL_BFLY2_OPERANDS);
#undef GL_BFLY2_OPERANDS
    s0 = pack64(s0l, s0h);
    d0 = pack64(d0l, d0h);
    s1 = pack64(s1l, s1h);
    d1 = pack64(d1l, d1h);
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

extern "C" __global__ __launch_bounds__(128) void k(const uint64_t* in, uint64_t* out, uint64_t stride) {
  const uint64_t* w = in + blockIdx.x*128 + threadIdx.x;
  uint64_t r0 = w[0*stride];
  uint64_t r1 = w[1*stride];
  uint64_t r2 = w[2*stride];
  uint64_t r3 = w[3*stride];
  uint64_t r4 = w[4*stride];
  uint64_t r5 = w[5*stride];
  uint64_t r6 = w[6*stride];
  uint64_t r7 = w[7*stride];
  uint64_t r8 = w[8*stride];
  uint64_t r9 = w[9*stride];
  uint64_t r10 = w[10*stride];
  uint64_t r11 = w[11*stride];
  uint64_t r12 = w[12*stride];
  uint64_t r13 = w[13*stride];
  uint64_t r14 = w[14*stride];
  uint64_t r15 = w[15*stride];
  uint64_t r16 = gl::sub(r13, r3);
  uint64_t r17 = gl::sub(r11, r2);
  uint64_t r18 = gl::sub(r5, r17);
  uint64_t r19 = gl::mul(r4, r7);
  uint64_t r20 = gl::mul(r11, r15);
  uint64_t r21 = gl::sub(r11, r7);
  uint64_t r22 = gl::sub(r5, r1);
  uint64_t r23 = gl::sub(r4, r14);
  uint64_t r24 = gl::mul(r22, r6);
  uint64_t r25 = gl::sub(r24, r17);
  uint64_t r26 = gl::mul(r22, r25);
  uint64_t r27 = gl::mul(r17, r5);
  uint64_t r28 = gl::add(r8, r5);
  uint64_t r29 = gl::add(r2, r0);
  uint64_t r30 = gl::add(r26, r0);
  uint64_t r31 = gl::mul(r9, r16);
  uint64_t r32 = gl::sub(r12, r18);
  uint64_t r33 = gl::mul(r20, r14);
  uint64_t r34 = gl::mul(r22, r31);
  uint64_t r35 = gl::sub(r18, r24);
  uint64_t r36 = gl::add(r23, r31);
  uint64_t r37 = gl::mul(r30, r21);
  uint64_t r38 = gl::sub(r3, r9);
  uint64_t r39 = gl::add(r4, r14);
  uint64_t r40 = gl::mul(r1, r23);
  uint64_t r41 = gl::add(r14, r9);
  uint64_t r42 = gl::add(r24, r23);
  uint64_t r43 = gl::sub(r33, r15);
  uint64_t r44 = gl::mul(r8, r18);
  uint64_t r45 = gl::sub(r24, r24);
  uint64_t r46 = gl::mul(r26, r6);
  uint64_t r47 = gl::sub(r13, r38);
  uint64_t r48 = gl::sub(r47, r37);
  uint64_t r49 = gl::mul(r34, r42);
  uint64_t r50 = gl::mul(r29, r13);
  uint64_t r51 = gl::mul(r23, r23);
  uint64_t r52 = gl::mul(r33, r24);
  uint64_t r53 = gl::mul(r19, r17);
  uint64_t r54 = gl::mul(r49, r31);
  uint64_t r55 = gl::mul(r50, r36);
  uint64_t r56 = gl::mul(r29, r19);
  uint64_t r57 = gl::mul(r26, r40);
  uint64_t r58 = gl::add(r23, r49);
  uint64_t r59 = gl::add(r33, r32);
  uint64_t r60 = gl::mul(r59, r21);
  uint64_t r61 = gl::sub(r36, r30);
  uint64_t r62 = gl::mul(r36, r52);
  uint64_t r63 = gl::sub(r39, r56);
  uint64_t r64 = gl::sub(r24, r26);
  uint64_t r65 = gl::sub(r63, r28);
  uint64_t r66 = gl::add(r30, r56);
  uint64_t r67 = gl::sub(r40, r28);
  uint64_t r68 = gl::mul(r34, r46);
  uint64_t r69 = gl::add(r50, r43);
  uint64_t r70 = gl::mul(r44, r30);
  uint64_t r71 = gl::mul(r68, r69);
  uint64_t r72 = gl::sub(r34, r66);
  uint64_t r73 = gl::mul(r63, r42);
  uint64_t r74 = gl::mul(r58, r61);
  uint64_t r75 = gl::mul(r73, r73);
  uint64_t r76 = gl::add(r57, r42);
  uint64_t r77 = gl::add(r57, r62);
  uint64_t r78 = gl::mul(r69, r65);
  uint64_t r79 = gl::add(r50, r48);
  uint64_t r80 = gl::mul(r67, r58);
  uint64_t r81 = gl::sub(r57, r49);
  uint64_t r82 = gl::sub(r67, r52);
  uint64_t r83 = gl::add(r57, r59);
  uint64_t r84 = gl::add(r78, r48);
  uint64_t r85 = gl::mul(r79, r78);
  uint64_t r86 = gl::sub(r57, r55);
  uint64_t r87 = gl::mul(r75, r79);
  uint64_t r88 = gl::sub(r60, r56);
  uint64_t r89 = gl::add(r76, r49);
  uint64_t r90 = gl::sub(r54, r55);
  uint64_t r91 = gl::mul(r84, r85);
  uint64_t r92 = gl::sub(r81, r75);
  uint64_t r93 = gl::sub(r88, r86);
  uint64_t r94 = gl::sub(r83, r54);
  uint64_t r95 = gl::add(r73, r83);
  uint64_t r96 = gl::mul(r68, r71);
  uint64_t r97 = gl::mul(r94, r59);
  uint64_t r98 = gl::mul(r74, r66);
  uint64_t r99 = gl::sub(r61, r73);
  uint64_t r100 = gl::sub(r92, r74);
  uint64_t r101 = gl::mul(r79, r80);
  uint64_t r102 = gl::sub(r66, r93);
  uint64_t r103 = gl::mul(r86, r78);
  uint64_t r104 = gl::mul(r67, r77);
  uint64_t r105 = gl::sub(r92, r94);
  uint64_t r106 = gl::add(r68, r86);
  uint64_t r107 = gl::add(r81, r79);
  uint64_t r108 = gl::mul(r75, r102);
  uint64_t r109 = gl::mul(r96, r101);
  uint64_t r110 = gl::add(r101, r79);
  uint64_t r111 = gl::sub(r78, r96);
  uint64_t r112 = gl::add(r84, r83);
  uint64_t r113 = gl::sub(r78, r73);
  uint64_t r114 = gl::mul(r95, r105);
  uint64_t r115 = gl::mul(r97, r106);
  uint64_t r116 = gl::sub(r86, r100);
  uint64_t r117 = gl::mul(r91, r88);
  uint64_t r118 = gl::add(r85, r101);
  uint64_t r119 = gl::add(r85, r115);
  uint64_t r120 = gl::mul(r101, r94);
  uint64_t r121 = gl::sub(r116, r108);
  uint64_t r122 = gl::add(r111, r114);
  uint64_t r123 = gl::sub(r119, r85);
  uint64_t r124 = gl::mul(r92, r87);
  uint64_t r125 = gl::mul(r105, r117);
  uint64_t r126 = gl::add(r88, r122);
  uint64_t r127 = gl::mul(r111, r109);
  uint64_t r128 = gl::add(r124, r103);
  uint64_t r129 = gl::mul(r99, r119);
  uint64_t r130 = gl::mul(r111, r101);
  uint64_t r131 = gl::mul(r105, r109);
  uint64_t r132 = gl::sub(r109, r106);
  uint64_t r133 = gl::mul(r107, r128);
  uint64_t r134 = gl::add(r132, r111);
  uint64_t r135 = gl::mul(r112, r132);
  uint64_t r136 = gl::mul(r135, r120);
  uint64_t r137 = gl::mul(r120, r120);
  uint64_t r138 = gl::add(r134, r118);
  uint64_t r139 = gl::mul(r137, r132);
  uint64_t r140 = gl::add(r139, r133);
  uint64_t r141 = gl::add(r138, r112);
  uint64_t r142 = gl::mul(r128, r121);
  uint64_t r143 = gl::add(r142, r109);
  uint64_t r144 = gl::add(r116, r140);
  uint64_t r145 = gl::mul(r119, r136);
  uint64_t r146 = gl::add(r136, r138);
  uint64_t r147 = gl::mul(r119, r125);
  uint64_t r148 = gl::add(r147, r120);
  uint64_t r149 = gl::add(r121, r115);
  uint64_t r150 = gl::mul(r117, r122);
  uint64_t r151 = gl::sub(r146, r117);
  uint64_t r152 = gl::mul(r151, r131);
  uint64_t r153 = gl::sub(r115, r140);
  uint64_t r154 = gl::mul(r148, r119);
  uint64_t r155 = gl::sub(r128, r117);
  uint64_t r156 = gl::sub(r126, r121);
  uint64_t r157 = gl::mul(r124, r146);
  uint64_t r158 = gl::add(r145, r150);
  uint64_t r159 = gl::add(r150, r133);
  uint64_t r160 = gl::mul(r155, r141);
  uint64_t r161 = gl::mul(r122, r150);
  uint64_t r162 = gl::sub(r146, r143);
  uint64_t r163 = gl::mul(r141, r153);
  uint64_t r164 = gl::mul(r152, r136);
  uint64_t r165 = gl::mul(r128, r134);
  uint64_t r166 = gl::mul(r131, r157);
  uint64_t r167 = gl::sub(r130, r153);
  uint64_t r168 = gl::add(r140, r153);
  uint64_t r169 = gl::mul(r146, r140);
  uint64_t r170 = gl::mul(r166, r143);
  uint64_t r171 = gl::sub(r145, r139);
  uint64_t r172 = gl::add(r164, r150);
  uint64_t r173 = gl::add(r141, r148);
  uint64_t r174 = gl::mul(r138, r163);
  uint64_t r175 = gl::sub(r166, r155);
  uint64_t r176 = gl::sub(r146, r154);
  uint64_t r177 = gl::sub(r166, r144);
  uint64_t r178 = gl::mul(r140, r144);
  uint64_t r179 = gl::mul(r142, r174);
  uint64_t r180 = gl::add(r146, r147);
  uint64_t r181 = gl::mul(r164, r154);
  uint64_t r182 = gl::mul(r171, r149);
  uint64_t r183 = gl::mul(r182, r165);
  uint64_t r184 = gl::sub(r155, r175);
  uint64_t r185 = gl::mul(r163, r180);
  uint64_t r186 = gl::mul(r151, r163);
  uint64_t r187 = gl::sub(r170, r168);
  uint64_t r188 = gl::add(r170, r148);
  uint64_t r189 = gl::mul(r181, r166);
  uint64_t r190 = gl::add(r177, r159);
  uint64_t r191 = gl::sub(r169, r166);
  uint64_t r192 = gl::mul(r160, r185);
  uint64_t r193 = gl::sub(r167, r163);
  uint64_t r194 = gl::mul(r174, r172);
  uint64_t r195 = gl::mul(r161, r184);
  uint64_t r196 = gl::add(r181, r190);
  uint64_t r197 = gl::sub(r157, r158);
  uint64_t r198 = gl::sub(r189, r191);
  uint64_t r199 = gl::add(r182, r193);
  uint64_t r200 = gl::mul(r172, r180);
  uint64_t r201 = gl::mul(r163, r175);
  uint64_t r202 = gl::sub(r190, r201);
  uint64_t r203 = gl::mul(r191, r201);
  uint64_t r204 = gl::add(r165, r170);
  uint64_t r205 = gl::mul(r168, r179);
  uint64_t r206 = gl::mul(r198, r186);
  uint64_t r207 = gl::mul(r206, r202);
  uint64_t r208 = gl::mul(r168, r201);
  uint64_t r209 = gl::mul(r205, r185);
  uint64_t r210 = gl::mul(r198, r200);
  uint64_t r211 = gl::mul(r190, r202);
  uint64_t r212 = gl::mul(r202, r188);
  uint64_t r213 = gl::mul(r187, r195);
  uint64_t r214 = gl::add(r188, r178);
  uint64_t r215 = gl::add(r205, r197);
  uint64_t r216 = gl::add(r214, r193);
  uint64_t r217 = gl::add(r182, r177);
  uint64_t r218 = gl::add(r215, r201);
  uint64_t r219 = gl::mul(r199, r198);
  uint64_t r220 = gl::sub(r211, r207);
  uint64_t r221 = gl::sub(r181, r197);
  uint64_t r222 = gl::sub(r212, r184);
  uint64_t r223 = gl::mul(r202, r220);
  uint64_t r224 = gl::add(r195, r185);
  uint64_t r225 = gl::add(r199, r195);
  uint64_t r226 = gl::mul(r204, r219);
  uint64_t r227 = gl::add(r210, r218);
  uint64_t r228 = gl::mul(r200, r196);
  uint64_t r229 = gl::mul(r203, r210);
  uint64_t r230 = gl::mul(r226, r211);
  uint64_t r231 = gl::sub(r224, r197);
  uint64_t r232 = gl::sub(r209, r218);
  uint64_t r233 = gl::mul(r228, r212);
  uint64_t r234 = gl::add(r211, r228);
  uint64_t r235 = gl::add(r212, r216);
  uint64_t r236 = gl::sub(r212, r210);
  uint64_t r237 = gl::sub(r232, r210);
  uint64_t r238 = gl::mul(r221, r217);
  uint64_t r239 = gl::mul(r226, r229);
  uint64_t r240 = gl::sub(r220, r228);
  uint64_t r241 = gl::mul(r235, r211);
  uint64_t r242 = gl::mul(r214, r218);
  uint64_t r243 = gl::sub(r205, r214);
  uint64_t r244 = gl::add(r235, r236);
  uint64_t r245 = gl::sub(r231, r244);
  uint64_t r246 = gl::mul(r206, r245);
  uint64_t r247 = gl::mul(r219, r221);
  uint64_t r248 = gl::mul(r240, r241);
  uint64_t r249 = gl::sub(r212, r243);
  uint64_t r250 = gl::sub(r235, r248);
  uint64_t r251 = gl::mul(r240, r243);
  uint64_t r252 = gl::mul(r234, r237);
  uint64_t r253 = gl::sub(r220, r227);
  uint64_t r254 = gl::add(r217, r237);
  uint64_t r255 = gl::sub(r222, r216);
  uint64_t r256 = gl::mul(r221, r254);
  uint64_t r257 = gl::mul(r248, r222);
  uint64_t r258 = gl::add(r251, r221);
  uint64_t r259 = gl::sub(r226, r236);
  uint64_t r260 = gl::mul(r230, r258);
  uint64_t r261 = gl::mul(r253, r221);
  uint64_t r262 = gl::add(r236, r250);
  uint64_t r263 = gl::add(r256, r241);
  uint64_t r264 = gl::sub(r224, r243);
  uint64_t r265 = gl::sub(r251, r226);
  uint64_t r266 = gl::mul(r241, r241);
  uint64_t r267 = gl::mul(r262, r237);
  uint64_t r268 = gl::sub(r235, r250);
  uint64_t r269 = gl::sub(r233, r265);
  uint64_t r270 = gl::sub(r268, r234);
  uint64_t r271 = gl::mul(r236, r247);
  uint64_t r272 = gl::mul(r249, r261);
  uint64_t r273 = gl::mul(r270, r260);
  uint64_t r274 = gl::sub(r248, r270);
  uint64_t r275 = gl::add(r266, r249);
  uint64_t r276 = gl::mul(r258, r237);
  uint64_t r277 = gl::sub(r268, r238);
  uint64_t r278 = gl::add(r246, r240);
  uint64_t r279 = gl::mul(r271, r265);
  uint64_t r280 = gl::mul(r242, r241);
  uint64_t r281 = gl::sub(r273, r270);
  uint64_t r282 = gl::mul(r249, r263);
  uint64_t r283 = gl::sub(r266, r272);
  uint64_t r284 = gl::sub(r250, r244);
  uint64_t r285 = gl::mul(r256, r254);
  uint64_t r286 = gl::mul(r278, r254);
  uint64_t r287 = gl::sub(r273, r250);
  uint64_t r288 = gl::sub(r254, r272);
  uint64_t r289 = gl::add(r253, r262);
  uint64_t r290 = gl::sub(r286, r255);
  uint64_t r291 = gl::mul(r270, r277);
  uint64_t r292 = gl::add(r254, r272);
  uint64_t r293 = gl::add(r290, r267);
  uint64_t r294 = gl::mul(r276, r293);
  uint64_t r295 = gl::mul(r275, r259);
  uint64_t r296 = gl::mul(r295, r289);
  uint64_t r297 = gl::mul(r273, r262);
  uint64_t r298 = gl::add(r281, r269);
  uint64_t r299 = gl::mul(r292, r283);
  uint64_t r300 = gl::sub(r283, r281);
  uint64_t r301 = gl::add(r276, r284);
  uint64_t r302 = gl::add(r272, r295);
  uint64_t r303 = gl::add(r284, r297);
  uint64_t r304 = gl::add(r288, r283);
  uint64_t r305 = gl::mul(r268, r301);
  uint64_t r306 = gl::add(r271, r290);
  uint64_t r307 = gl::sub(r291, r278);
  uint64_t r308 = gl::sub(r279, r292);
  uint64_t r309 = gl::add(r305, r286);
  uint64_t r310 = gl::mul(r290, r282);
  uint64_t r311 = gl::mul(r286, r302);
  uint64_t r312 = gl::mul(r309, r302);
  uint64_t r313 = gl::mul(r283, r287);
  uint64_t r314 = gl::mul(r277, r303);
  uint64_t r315 = gl::sub(r278, r287);
  uint64_t r316 = gl::mul(r296, r301);
  uint64_t r317 = gl::mul(r290, r308);
  uint64_t r318 = gl::add(r295, r303);
  uint64_t r319 = gl::sub(r315, r306);
  uint64_t r320 = gl::sub(r303, r309);
  uint64_t r321 = gl::mul(r288, r285);
  uint64_t r322 = gl::add(r301, r285);
  uint64_t r323 = gl::add(r318, r303);
  uint64_t r324 = gl::mul(r309, r312);
  uint64_t r325 = gl::sub(r287, r301);
  uint64_t r326 = gl::mul(r320, r323);
  uint64_t r327 = gl::mul(r299, r305);
  uint64_t r328 = gl::sub(r296, r299);
  uint64_t r329 = gl::sub(r313, r294);
  uint64_t r330 = gl::mul(r302, r323);
  uint64_t r331 = gl::sub(r307, r295);
  uint64_t r332 = gl::mul(r318, r299);
  uint64_t r333 = gl::mul(r309, r301);
  uint64_t r334 = gl::mul(r302, r308);
  uint64_t r335 = gl::add(r325, r303);
  uint64_t r336 = gl::mul(r311, r302);
  uint64_t r337 = gl::mul(r320, r320);
  uint64_t r338 = gl::sub(r301, r316);
  uint64_t r339 = gl::add(r320, r303);
  uint64_t r340 = gl::mul(r315, r336);
  uint64_t r341 = gl::add(r339, r307);
  uint64_t r342 = gl::add(r309, r341);
  uint64_t r343 = gl::mul(r305, r318);
  uint64_t r344 = gl::add(r325, r330);
  uint64_t r345 = gl::sub(r325, r314);
  uint64_t r346 = gl::mul(r339, r314);
  uint64_t r347 = gl::add(r334, r320);
  uint64_t r348 = gl::sub(r319, r330);
  uint64_t r349 = gl::mul(r330, r330);
  uint64_t r350 = gl::mul(r347, r323);
  uint64_t r351 = gl::sub(r333, r316);
  uint64_t r352 = gl::sub(r332, r341);
  uint64_t r353 = gl::sub(r326, r346);
  uint64_t r354 = gl::sub(r343, r331);
  uint64_t r355 = gl::sub(r317, r351);
  uint64_t r356 = gl::sub(r329, r321);
  uint64_t r357 = gl::sub(r352, r328);
  uint64_t r358 = gl::add(r338, r337);
  uint64_t r359 = gl::sub(r358, r357);
  uint64_t r360 = gl::sub(r347, r355);
  uint64_t r361 = gl::mul(r346, r358);
  uint64_t r362 = gl::mul(r342, r347);
  uint64_t r363 = gl::add(r346, r330);
  uint64_t r364 = gl::mul(r338, r324);
  uint64_t r365 = gl::mul(r343, r341);
  uint64_t r366 = gl::mul(r346, r358);
  uint64_t r367 = gl::mul(r330, r363);
  uint64_t r368 = gl::sub(r354, r341);
  uint64_t r369 = gl::sub(r335, r345);
  uint64_t r370 = gl::mul(r355, r367);
  uint64_t r371 = gl::sub(r338, r365);
  uint64_t r372 = gl::add(r333, r347);
  uint64_t r373 = gl::mul(r357, r339);
  uint64_t r374 = gl::add(r361, r350);
  uint64_t r375 = gl::sub(r351, r339);
  uint64_t r376 = gl::add(r344, r367);
  uint64_t r377 = gl::add(r358, r351);
  uint64_t r378 = gl::add(r347, r338);
  uint64_t r379 = gl::mul(r371, r348);
  uint64_t r380 = gl::mul(r355, r353);
  uint64_t r381 = gl::add(r376, r358);
  uint64_t r382 = gl::mul(r354, r347);
  uint64_t r383 = gl::mul(r357, r355);
  uint64_t r384 = gl::mul(r351, r347);
  uint64_t r385 = gl::sub(r362, r364);
  uint64_t r386 = gl::sub(r351, r373);
  uint64_t r387 = gl::mul(r382, r354);
  uint64_t r388 = gl::add(r385, r375);
  uint64_t r389 = gl::mul(r371, r376);
  uint64_t r390 = gl::mul(r377, r370);
  uint64_t r391 = gl::add(r353, r374);
  uint64_t r392 = gl::sub(r386, r379);
  uint64_t r393 = gl::add(r377, r384);
  uint64_t r394 = gl::mul(r363, r363);
  uint64_t r395 = gl::add(r381, r359);
  uint64_t r396 = gl::mul(r377, r372);
  uint64_t r397 = gl::sub(r394, r393);
  uint64_t r398 = gl::mul(r389, r390);
  uint64_t r399 = gl::sub(r360, r392);
  uint64_t r400 = gl::sub(r374, r364);
  uint64_t r401 = gl::mul(r387, r375);
  uint64_t r402 = gl::add(r370, r373);
  uint64_t r403 = gl::mul(r368, r393);
  uint64_t r404 = gl::sub(r375, r395);
  uint64_t r405 = gl::mul(r400, r394);
  uint64_t r406 = gl::sub(r395, r382);
  uint64_t r407 = gl::mul(r405, r404);
  uint64_t r408 = gl::mul(r374, r388);
  uint64_t r409 = gl::mul(r407, r375);
  uint64_t r410 = gl::add(r393, r403);
  uint64_t r411 = gl::add(r383, r406);
  uint64_t r412 = gl::sub(r386, r374);
  uint64_t r413 = gl::mul(r392, r375);
  uint64_t r414 = gl::add(r388, r410);
  uint64_t r415 = gl::mul(r392, r389);
  uint64_t r416 = gl::sub(r406, r377);
  uint64_t r417 = gl::mul(r384, r397);
  uint64_t r418 = gl::mul(r390, r410);
  uint64_t r419 = gl::mul(r383, r398);
  uint64_t r420 = gl::mul(r396, r412);
  uint64_t r421 = gl::sub(r386, r391);
  uint64_t r422 = gl::add(r416, r394);
  uint64_t r423 = gl::mul(r402, r395);
  uint64_t r424 = gl::sub(r423, r417);
  uint64_t r425 = gl::mul(r407, r385);
  uint64_t r426 = gl::add(r425, r422);
  uint64_t r427 = gl::sub(r388, r400);
  uint64_t r428 = gl::sub(r392, r396);
  uint64_t r429 = gl::add(r405, r408);
  uint64_t r430 = gl::mul(r413, r394);
  uint64_t r431 = gl::mul(r420, r417);
  uint64_t r432 = gl::sub(r398, r401);
  uint64_t r433 = gl::mul(r421, r425);
  uint64_t r434 = gl::mul(r427, r423);
  uint64_t r435 = gl::mul(r422, r401);
  uint64_t r436 = gl::mul(r414, r431);
  uint64_t r437 = gl::sub(r401, r411);
  uint64_t r438 = gl::sub(r410, r430);
  uint64_t r439 = gl::sub(r429, r403);
  uint64_t r440 = gl::sub(r402, r425);
  uint64_t r441 = gl::sub(r429, r404);
  uint64_t r442 = gl::sub(r441, r418);
  uint64_t r443 = gl::mul(r418, r429);
  uint64_t r444 = gl::mul(r426, r433);
  uint64_t r445 = gl::add(r406, r426);
  uint64_t r446 = gl::mul(r438, r424);
  uint64_t r447 = gl::mul(r430, r421);
  uint64_t r448 = gl::mul(r434, r447);
  uint64_t r449 = gl::mul(r409, r447);
  uint64_t r450 = gl::mul(r437, r439);
  uint64_t r451 = gl::mul(r442, r448);
  uint64_t r452 = gl::mul(r418, r438);
  uint64_t r453 = gl::mul(r413, r442);
  uint64_t r454 = gl::mul(r428, r435);
  uint64_t r455 = gl::sub(r432, r421);
  uint64_t r456 = gl::add(r419, r418);
  uint64_t r457 = gl::sub(r449, r452);
  uint64_t r458 = gl::mul(r450, r426);
  uint64_t r459 = gl::mul(r448, r421);
  uint64_t r460 = gl::mul(r440, r448);
  uint64_t r461 = gl::add(r438, r448);
  uint64_t r462 = gl::mul(r427, r431);
  uint64_t r463 = gl::add(r447, r449);
  uint64_t r464 = gl::mul(r436, r449);
  uint64_t r465 = gl::mul(r452, r440);
  uint64_t r466 = gl::add(r440, r445);
  uint64_t r467 = gl::add(r461, r461);
  uint64_t r468 = gl::sub(r453, r434);
  uint64_t r469 = gl::sub(r467, r432);
  uint64_t r470 = gl::sub(r433, r463);
  uint64_t r471 = gl::mul(r437, r458);
  uint64_t r472 = gl::add(r444, r436);
  uint64_t r473 = gl::sub(r450, r434);
  uint64_t r474 = gl::add(r464, r438);
  uint64_t r475 = gl::mul(r438, r450);
  uint64_t r476 = gl::mul(r464, r464);
  uint64_t r477 = gl::mul(r465, r466);
  uint64_t r478 = gl::mul(r452, r442);
  uint64_t r479 = gl::mul(r453, r459);
  uint64_t r480 = gl::add(r459, r467);
  uint64_t r481 = gl::sub(r475, r444);
  uint64_t r482 = gl::add(r472, r472);
  uint64_t r483 = gl::mul(r453, r454);
  uint64_t r484 = gl::mul(r478, r446);
  uint64_t r485 = gl::mul(r447, r482);
  uint64_t r486 = gl::sub(r477, r452);
  uint64_t r487 = gl::mul(r468, r467);
  uint64_t r488 = gl::add(r455, r457);
  uint64_t r489 = gl::mul(r467, r459);
  uint64_t r490 = gl::sub(r464, r489);
  uint64_t r491 = gl::sub(r453, r482);
  uint64_t r492 = gl::mul(r468, r468);
  uint64_t r493 = gl::add(r468, r484);
  uint64_t r494 = gl::mul(r489, r480);
  uint64_t r495 = gl::mul(r458, r473);
  uint64_t r496 = gl::add(r478, r480);
  uint64_t r497 = gl::mul(r469, r479);
  uint64_t r498 = gl::add(r479, r462);
  uint64_t r499 = gl::mul(r471, r469);
  uint64_t r500 = gl::sub(r477, r484);
  uint64_t r501 = gl::add(r467, r470);
  uint64_t r502 = gl::add(r467, r474);
  uint64_t r503 = gl::sub(r501, r471);
  uint64_t r504 = gl::sub(r477, r466);
  uint64_t r505 = gl::mul(r481, r468);
  uint64_t r506 = gl::sub(r475, r484);
  uint64_t r507 = gl::sub(r487, r504);
  uint64_t r508 = gl::mul(r476, r505);
  uint64_t r509 = gl::add(r482, r479);
  uint64_t r510 = gl::add(r497, r508);
  uint64_t r511 = gl::mul(r495, r480);
  uint64_t r512 = gl::mul(r485, r501);
  uint64_t r513 = gl::sub(r501, r474);
  uint64_t r514 = gl::sub(r484, r481);
  uint64_t r515 = gl::mul(r509, r487);
  uint64_t r516 = gl::mul(r490, r489);
  uint64_t r517 = gl::add(r514, r511);
  uint64_t r518 = gl::mul(r508, r483);
  uint64_t r519 = gl::add(r508, r482);
  uint64_t r520 = gl::sub(r500, r489);
  uint64_t r521 = gl::sub(r481, r509);
  uint64_t r522 = gl::add(r488, r521);
  uint64_t r523 = gl::mul(r497, r513);
  uint64_t r524 = gl::mul(r484, r485);
  uint64_t r525 = gl::mul(r500, r506);
  uint64_t r526 = gl::sub(r517, r495);
  uint64_t r527 = gl::mul(r514, r500);
  uint64_t r528 = gl::mul(r501, r504);
  uint64_t r529 = gl::mul(r513, r495);
  uint64_t r530 = gl::add(r505, r493);
  uint64_t r531 = gl::mul(r509, r508);
  uint64_t r532 = gl::mul(r524, r500);
  uint64_t r533 = gl::sub(r522, r523);
  uint64_t r534 = gl::mul(r513, r510);
  uint64_t r535 = gl::mul(r503, r500);
  uint64_t r536 = gl::add(r524, r498);
  uint64_t r537 = gl::add(r512, r501);
  uint64_t r538 = gl::mul(r506, r517);
  uint64_t r539 = gl::mul(r523, r533);
  uint64_t r540 = gl::sub(r527, r524);
  uint64_t r541 = gl::mul(r522, r523);
  uint64_t r542 = gl::sub(r537, r506);
  uint64_t r543 = gl::mul(r524, r526);
  uint64_t r544 = gl::add(r505, r511);
  uint64_t r545 = gl::mul(r543, r536);
  uint64_t r546 = gl::mul(r540, r514);
  uint64_t r547 = gl::add(r537, r541);
  uint64_t r548 = gl::sub(r521, r545);
  uint64_t r549 = gl::add(r526, r521);
  uint64_t r550 = gl::mul(r516, r541);
  uint64_t r551 = gl::mul(r539, r544);
  uint64_t r552 = gl::mul(r522, r517);
  uint64_t r553 = gl::mul(r542, r550);
  uint64_t r554 = gl::mul(r553, r525);
  uint64_t r555 = gl::add(r544, r552);
  uint64_t r556 = gl::add(r525, r520);
  uint64_t r557 = gl::sub(r551, r523);
  uint64_t r558 = gl::sub(r553, r526);
  uint64_t r559 = gl::mul(r528, r531);
  uint64_t r560 = gl::mul(r549, r546);
  uint64_t r561 = gl::mul(r547, r530);
  uint64_t r562 = gl::sub(r557, r554);
  uint64_t r563 = gl::mul(r553, r539);
  uint64_t r564 = gl::sub(r541, r539);
  uint64_t r565 = gl::mul(r539, r528);
  uint64_t r566 = gl::mul(r531, r544);
  uint64_t r567 = gl::add(r549, r559);
  uint64_t r568 = gl::mul(r547, r546);
  uint64_t r569 = gl::sub(r544, r554);
  uint64_t r570 = gl::mul(r560, r545);
  uint64_t r571 = gl::add(r556, r557);
  uint64_t r572 = gl::sub(r567, r556);
  uint64_t r573 = gl::mul(r546, r553);
  uint64_t r574 = gl::sub(r563, r568);
  uint64_t r575 = gl::add(r574, r546);
  uint64_t r576 = gl::mul(r538, r556);
  uint64_t r577 = gl::mul(r537, r545);
  uint64_t r578 = gl::mul(r552, r568);
  uint64_t r579 = gl::mul(r571, r543);
  uint64_t r580 = gl::sub(r555, r544);
  uint64_t r581 = gl::mul(r576, r573);
  uint64_t r582 = gl::mul(r552, r580);
  uint64_t r583 = gl::add(r565, r552);
  uint64_t r584 = gl::mul(r574, r545);
  uint64_t r585 = gl::mul(r550, r570);
  uint64_t r586 = gl::mul(r569, r576);
  uint64_t r587 = gl::sub(r547, r563);
  uint64_t r588 = gl::sub(r568, r552);
  uint64_t r589 = gl::mul(r576, r585);
  uint64_t r590 = gl::add(r587, r561);
  uint64_t r591 = gl::mul(r563, r589);
  uint64_t r592 = gl::sub(r582, r560);
  uint64_t r593 = gl::mul(r578, r583);
  uint64_t r594 = gl::add(r573, r590);
  uint64_t r595 = gl::add(r566, r563);
  uint64_t r596 = gl::mul(r575, r557);
  uint64_t r597 = gl::sub(r584, r593);
  uint64_t r598 = gl::mul(r560, r569);
  uint64_t r599 = gl::sub(r569, r559);
  uint64_t r600 = gl::mul(r569, r585);
  uint64_t r601 = gl::mul(r573, r563);
  uint64_t r602 = gl::sub(r585, r590);
  uint64_t r603 = gl::mul(r582, r571);
  uint64_t r604 = gl::sub(r598, r582);
  uint64_t r605 = gl::add(r584, r599);
  uint64_t r606 = gl::add(r572, r567);
  uint64_t r607 = gl::sub(r591, r598);
  uint64_t r608 = gl::mul(r591, r586);
  uint64_t r609 = gl::sub(r599, r594);
  uint64_t r610 = gl::mul(r580, r582);
  uint64_t r611 = gl::mul(r589, r573);
  uint64_t r612 = gl::add(r585, r583);
  uint64_t r613 = gl::mul(r598, r581);
  uint64_t r614 = gl::sub(r608, r587);
  uint64_t r615 = gl::sub(r602, r611);
  uint64_t r616 = gl::add(r587, r600);
  uint64_t r617 = gl::mul(r586, r609);
  uint64_t r618 = gl::mul(r588, r591);
  uint64_t r619 = gl::mul(r597, r588);
  uint64_t r620 = gl::add(r593, r605);
  uint64_t r621 = gl::add(r582, r584);
  uint64_t r622 = gl::mul(r597, r589);
  uint64_t r623 = gl::sub(r583, r605);
  uint64_t r624 = gl::mul(r612, r593);
  uint64_t r625 = gl::mul(r601, r603);
  uint64_t r626 = gl::mul(r590, r619);
  uint64_t r627 = gl::sub(r595, r595);
  uint64_t r628 = gl::mul(r594, r616);
  uint64_t r629 = gl::sub(r600, r607);
  uint64_t r630 = gl::sub(r598, r618);
  uint64_t r631 = gl::sub(r613, r626);
  uint64_t r632 = gl::add(r600, r593);
  uint64_t r633 = gl::sub(r614, r613);
  uint64_t r634 = gl::mul(r598, r600);
  uint64_t r635 = gl::mul(r628, r605);
  uint64_t r636 = gl::sub(r597, r616);
  uint64_t r637 = gl::add(r598, r624);
  uint64_t r638 = gl::add(r612, r607);
  uint64_t r639 = gl::sub(r637, r605);
  uint64_t r640 = gl::mul(r601, r630);
  uint64_t r641 = gl::add(r622, r625);
  uint64_t r642 = gl::mul(r625, r614);
  uint64_t r643 = gl::mul(r634, r618);
  uint64_t r644 = gl::add(r611, r623);
  uint64_t r645 = gl::mul(r640, r606);
  uint64_t r646 = gl::mul(r614, r618);
  uint64_t r647 = gl::sub(r629, r636);
  uint64_t r648 = gl::sub(r623, r618);
  uint64_t r649 = gl::mul(r640, r616);
  uint64_t r650 = gl::mul(r631, r613);
  uint64_t r651 = gl::sub(r613, r621);
  uint64_t r652 = gl::mul(r631, r647);
  uint64_t r653 = gl::sub(r625, r642);
  uint64_t r654 = gl::mul(r625, r647);
  uint64_t r655 = gl::sub(r632, r648);
  uint64_t r656 = gl::add(r628, r616);
  uint64_t r657 = gl::add(r626, r645);
  uint64_t r658 = gl::add(r635, r652);
  uint64_t r659 = gl::mul(r653, r635);
  uint64_t r660 = gl::mul(r630, r649);
  uint64_t r661 = gl::sub(r633, r638);
  uint64_t r662 = gl::mul(r653, r630);
  uint64_t r663 = gl::mul(r650, r625);
  uint64_t r664 = gl::add(r639, r655);
  uint64_t r665 = gl::add(r661, r660);
  uint64_t r666 = gl::mul(r629, r639);
  uint64_t r667 = gl::sub(r649, r628);
  uint64_t r668 = gl::add(r662, r663);
  uint64_t r669 = gl::sub(r640, r636);
  uint64_t r670 = gl::mul(r654, r655);
  uint64_t r671 = gl::mul(r632, r657);
  uint64_t r672 = gl::add(r657, r657);
  uint64_t r673 = gl::add(r647, r658);
  uint64_t r674 = gl::mul(r650, r670);
  uint64_t r675 = gl::mul(r667, r656);
  uint64_t r676 = gl::mul(r670, r657);
  uint64_t r677 = gl::mul(r646, r644);
  uint64_t r678 = gl::mul(r665, r663);
  uint64_t r679 = gl::sub(r667, r663);
  uint64_t r680 = gl::mul(r657, r663);
  uint64_t r681 = gl::mul(r646, r663);
  uint64_t r682 = gl::mul(r670, r647);
  uint64_t r683 = gl::sub(r656, r672);
  uint64_t r684 = gl::add(r673, r659);
  uint64_t r685 = gl::mul(r657, r660);
  uint64_t r686 = gl::add(r679, r680);
  uint64_t r687 = gl::mul(r662, r684);
  uint64_t r688 = gl::sub(r678, r655);
  uint64_t r689 = gl::mul(r676, r677);
  uint64_t r690 = gl::add(r659, r654);
  uint64_t r691 = gl::sub(r661, r674);
  uint64_t r692 = gl::add(r690, r652);
  uint64_t r693 = gl::mul(r692, r657);
  uint64_t r694 = gl::mul(r654, r677);
  uint64_t r695 = gl::sub(r660, r671);
  uint64_t r696 = gl::mul(r680, r666);
  uint64_t r697 = gl::sub(r683, r666);
  uint64_t r698 = gl::mul(r677, r690);
  uint64_t r699 = gl::mul(r680, r683);
  uint64_t r700 = gl::add(r676, r686);
  uint64_t r701 = gl::mul(r664, r678);
  uint64_t r702 = gl::mul(r684, r669);
  uint64_t r703 = gl::add(r668, r667);
  uint64_t r704 = gl::mul(r684, r679);
  uint64_t r705 = gl::mul(r699, r689);
  uint64_t r706 = gl::sub(r692, r698);
  uint64_t r707 = gl::sub(r679, r689);
  uint64_t r708 = gl::mul(r688, r696);
  uint64_t r709 = gl::mul(r669, r681);
  uint64_t r710 = gl::sub(r688, r690);
  uint64_t r711 = gl::add(r688, r680);
  uint64_t r712 = gl::add(r673, r673);
  uint64_t r713 = gl::mul(r693, r673);
  uint64_t r714 = gl::add(r711, r692);
  uint64_t r715 = gl::add(r705, r682);
  uint64_t r716 = gl::mul(r694, r699);
  uint64_t r717 = gl::sub(r706, r709);
  uint64_t r718 = gl::mul(r682, r681);
  uint64_t r719 = gl::mul(r704, r711);
  uint64_t r720 = gl::mul(r693, r694);
  uint64_t r721 = gl::add(r686, r717);
  uint64_t r722 = gl::mul(r684, r694);
  uint64_t r723 = gl::mul(r705, r698);
  uint64_t r724 = gl::mul(r691, r714);
  uint64_t r725 = gl::sub(r704, r698);
  uint64_t r726 = gl::mul(r708, r721);
  uint64_t r727 = gl::sub(r691, r719);
  uint64_t r728 = gl::add(r702, r702);
  uint64_t r729 = gl::mul(r726, r702);
  uint64_t r730 = gl::mul(r723, r725);
  uint64_t r731 = gl::mul(r704, r730);
  uint64_t r732 = gl::add(r698, r695);
  uint64_t r733 = gl::sub(r713, r702);
  uint64_t r734 = gl::mul(r720, r700);
  uint64_t r735 = gl::add(r700, r709);
  uint64_t r736 = gl::mul(r714, r721);
  uint64_t r737 = gl::add(r732, r720);
  uint64_t r738 = gl::mul(r729, r701);
  uint64_t r739 = gl::mul(r726, r711);
  uint64_t r740 = gl::mul(r737, r722);
  uint64_t r741 = gl::sub(r736, r701);
  uint64_t r742 = gl::add(r736, r724);
  uint64_t r743 = gl::mul(r726, r708);
  uint64_t r744 = gl::mul(r717, r708);
  uint64_t r745 = gl::mul(r733, r744);
  uint64_t r746 = gl::add(r722, r735);
  uint64_t r747 = gl::mul(r729, r716);
  uint64_t r748 = gl::mul(r738, r732);
  uint64_t r749 = gl::mul(r744, r713);
  uint64_t r750 = gl::mul(r738, r727);
  uint64_t r751 = gl::add(r742, r711);
  uint64_t r752 = gl::sub(r736, r745);
  uint64_t r753 = gl::add(r719, r752);
  uint64_t r754 = gl::add(r717, r723);
  uint64_t r755 = gl::mul(r744, r745);
  uint64_t r756 = gl::mul(r741, r733);
  uint64_t r757 = gl::mul(r745, r727);
  uint64_t r758 = gl::add(r729, r735);
  uint64_t r759 = gl::sub(r743, r727);
  uint64_t r760 = gl::sub(r729, r750);
  uint64_t r761 = gl::mul(r734, r730);
  uint64_t r762 = gl::mul(r745, r746);
  uint64_t r763 = gl::add(r735, r734);
  uint64_t r764 = gl::mul(r724, r753);
  uint64_t r765 = gl::mul(r751, r741);
  uint64_t r766 = gl::mul(r762, r729);
  uint64_t r767 = gl::mul(r738, r753);
  uint64_t r768 = gl::add(r744, r740);
  uint64_t r769 = gl::mul(r732, r743);
  uint64_t r770 = gl::mul(r762, r762);
  uint64_t r771 = gl::mul(r742, r753);
  uint64_t r772 = gl::add(r763, r746);
  uint64_t r773 = gl::sub(r737, r760);
  uint64_t r774 = gl::sub(r745, r758);
  uint64_t r775 = gl::mul(r746, r762);
  uint64_t r776 = gl::sub(r761, r746);
  uint64_t r777 = gl::mul(r743, r757);
  uint64_t r778 = gl::sub(r765, r745);
  uint64_t r779 = gl::add(r753, r765);
  uint64_t r780 = gl::mul(r750, r745);
  uint64_t r781 = gl::add(r772, r765);
  uint64_t r782 = gl::mul(r781, r759);
  uint64_t r783 = gl::add(r752, r756);
  uint64_t r784 = gl::mul(r746, r769);
  uint64_t r785 = gl::mul(r780, r779);
  uint64_t r786 = gl::add(r752, r776);
  uint64_t r787 = gl::add(r758, r758);
  uint64_t r788 = gl::mul(r781, r784);
  uint64_t r789 = gl::mul(r787, r761);
  uint64_t r790 = gl::add(r788, r751);
  uint64_t r791 = gl::add(r753, r786);
  uint64_t r792 = gl::mul(r774, r784);
  uint64_t r793 = gl::sub(r784, r768);
  uint64_t r794 = gl::sub(r762, r756);
  uint64_t r795 = gl::add(r786, r769);
  uint64_t r796 = gl::sub(r756, r771);
  uint64_t r797 = gl::sub(r774, r795);
  uint64_t r798 = gl::mul(r760, r783);
  uint64_t r799 = gl::mul(r770, r778);
  uint64_t r800 = gl::mul(r789, r793);
  uint64_t r801 = gl::mul(r778, r778);
  uint64_t r802 = gl::add(r779, r780);
  uint64_t r803 = gl::sub(r769, r772);
  uint64_t r804 = gl::add(r789, r781);
  uint64_t r805 = gl::mul(r767, r788);
  uint64_t r806 = gl::mul(r796, r793);
  uint64_t r807 = gl::add(r768, r775);
  uint64_t r808 = gl::mul(r807, r787);
  uint64_t r809 = gl::mul(r787, r781);
  uint64_t r810 = gl::mul(r793, r807);
  uint64_t r811 = gl::mul(r795, r798);
  uint64_t r812 = gl::sub(r785, r787);
  uint64_t r813 = gl::mul(r787, r775);
  uint64_t r814 = gl::sub(r812, r782);
  uint64_t r815 = gl::add(r785, r806);
  uint64_t r816 = gl::sub(r808, r805);
  uint64_t r817 = gl::mul(r812, r782);
  uint64_t r818 = gl::mul(r804, r810);
  uint64_t r819 = gl::mul(r785, r779);
  uint64_t r820 = gl::sub(r805, r803);
  uint64_t r821 = gl::mul(r802, r811);
  uint64_t r822 = gl::mul(r795, r783);
  uint64_t r823 = gl::add(r801, r811);
  uint64_t r824 = gl::sub(r811, r795);
  uint64_t r825 = gl::mul(r810, r800);
  uint64_t r826 = gl::add(r792, r795);
  uint64_t r827 = gl::add(r799, r823);
  uint64_t r828 = gl::mul(r818, r816);
  uint64_t r829 = gl::sub(r794, r820);
  uint64_t r830 = gl::sub(r816, r813);
  uint64_t r831 = gl::add(r812, r822);
  uint64_t r832 = gl::sub(r805, r803);
  uint64_t r833 = gl::mul(r823, r811);
  uint64_t r834 = gl::mul(r832, r809);
  uint64_t r835 = gl::sub(r832, r803);
  uint64_t r836 = gl::add(r796, r796);
  uint64_t r837 = gl::add(r834, r828);
  uint64_t r838 = gl::add(r824, r824);
  uint64_t r839 = gl::add(r836, r829);
  uint64_t r840 = gl::mul(r819, r825);
  uint64_t r841 = gl::mul(r819, r817);
  uint64_t r842 = gl::add(r819, r840);
  uint64_t r843 = gl::mul(r837, r834);
  uint64_t r844 = gl::mul(r805, r813);
  uint64_t r845 = gl::add(r822, r839);
  uint64_t r846 = gl::mul(r818, r830);
  uint64_t r847 = gl::add(r825, r836);
  uint64_t r848 = gl::sub(r819, r810);
  uint64_t r849 = gl::mul(r845, r810);
  uint64_t r850 = gl::add(r844, r810);
  uint64_t r851 = gl::sub(r815, r842);
  uint64_t r852 = gl::mul(r840, r837);
  uint64_t r853 = gl::mul(r836, r850);
  uint64_t r854 = gl::mul(r848, r851);
  uint64_t r855 = gl::sub(r826, r848);
  uint64_t r856 = gl::mul(r825, r826);
  uint64_t r857 = gl::mul(r849, r845);
  uint64_t r858 = gl::add(r831, r852);
  uint64_t r859 = gl::mul(r833, r847);
  uint64_t r860 = gl::sub(r831, r846);
  uint64_t r861 = gl::add(r833, r859);
  uint64_t r862 = gl::sub(r859, r843);
  uint64_t r863 = gl::mul(r827, r828);
  uint64_t r864 = gl::sub(r845, r828);
  uint64_t r865 = gl::mul(r829, r843);
  uint64_t r866 = gl::mul(r842, r855);
  uint64_t r867 = gl::mul(r844, r843);
  uint64_t r868 = gl::add(r844, r849);
  uint64_t r869 = gl::mul(r848, r864);
  uint64_t r870 = gl::sub(r839, r836);
  uint64_t r871 = gl::mul(r840, r834);
  uint64_t r872 = gl::mul(r834, r847);
  uint64_t r873 = gl::mul(r853, r844);
  uint64_t r874 = gl::mul(r853, r847);
  uint64_t r875 = gl::add(r838, r869);
  uint64_t r876 = gl::sub(r868, r875);
  uint64_t r877 = gl::sub(r871, r869);
  uint64_t r878 = gl::mul(r859, r867);
  uint64_t r879 = gl::mul(r865, r860);
  uint64_t r880 = gl::mul(r867, r867);
  uint64_t r881 = gl::mul(r870, r861);
  uint64_t r882 = gl::mul(r854, r869);
  uint64_t r883 = gl::sub(r856, r876);
  uint64_t r884 = gl::mul(r854, r867);
  uint64_t r885 = gl::add(r874, r861);
  uint64_t r886 = gl::add(r876, r871);
  uint64_t r887 = gl::mul(r881, r870);
  uint64_t r888 = gl::mul(r873, r863);
  uint64_t r889 = gl::add(r864, r852);
  uint64_t r890 = gl::add(r856, r850);
  uint64_t r891 = gl::mul(r890, r873);
  uint64_t r892 = gl::mul(r861, r865);
  uint64_t r893 = gl::mul(r874, r869);
  uint64_t r894 = gl::sub(r882, r881);
  uint64_t r895 = gl::mul(r881, r882);
  uint64_t r896 = gl::add(r888, r870);
  uint64_t r897 = gl::mul(r861, r891);
  uint64_t r898 = gl::mul(r895, r860);
  uint64_t r899 = gl::mul(r876, r863);
  uint64_t r900 = gl::add(r876, r889);
  uint64_t r901 = gl::mul(r881, r861);
  uint64_t r902 = gl::add(r877, r885);
  uint64_t r903 = gl::mul(r881, r899);
  uint64_t r904 = gl::sub(r878, r869);
  uint64_t r905 = gl::mul(r893, r890);
  uint64_t r906 = gl::sub(r872, r885);
  uint64_t r907 = gl::add(r886, r896);
  uint64_t r908 = gl::sub(r893, r895);
  uint64_t r909 = gl::mul(r900, r898);
  uint64_t r910 = gl::add(r884, r909);
  uint64_t r911 = gl::mul(r873, r885);
  uint64_t r912 = gl::mul(r898, r905);
  uint64_t r913 = gl::sub(r882, r884);
  uint64_t r914 = gl::sub(r874, r875);
  uint64_t r915 = gl::sub(r883, r895);
  uint64_t r916 = gl::mul(r896, r880);
  uint64_t r917 = gl::mul(r905, r891);
  uint64_t r918 = gl::sub(r902, r905);
  uint64_t r919 = gl::sub(r885, r907);
  uint64_t r920 = gl::mul(r905, r887);
  uint64_t r921 = gl::mul(r907, r911);
  uint64_t r922 = gl::sub(r891, r888);
  uint64_t r923 = gl::sub(r893, r885);
  uint64_t r924 = gl::mul(r893, r889);
  uint64_t r925 = gl::sub(r921, r889);
  uint64_t r926 = gl::mul(r902, r913);
  uint64_t r927 = gl::sub(r914, r904);
  uint64_t r928 = gl::add(r900, r897);
  uint64_t r929 = gl::add(r899, r909);
  uint64_t r930 = gl::sub(r905, r896);
  uint64_t r931 = gl::mul(r922, r898);
  uint64_t r932 = gl::mul(r926, r899);
  uint64_t r933 = gl::add(r913, r925);
  uint64_t r934 = gl::sub(r929, r896);
  uint64_t r935 = gl::mul(r921, r915);
  uint64_t r936 = gl::add(r931, r899);
  uint64_t r937 = gl::sub(r927, r923);
  uint64_t r938 = gl::mul(r899, r929);
  uint64_t r939 = gl::add(r916, r902);
  uint64_t r940 = gl::sub(r912, r923);
  uint64_t r941 = gl::sub(r920, r913);
  uint64_t r942 = gl::mul(r920, r923);
  uint64_t r943 = gl::sub(r940, r922);
  uint64_t r944 = gl::mul(r930, r912);
  uint64_t r945 = gl::mul(r942, r906);
  uint64_t r946 = gl::sub(r926, r935);
  uint64_t r947 = gl::add(r918, r907);
  uint64_t r948 = gl::add(r932, r923);
  uint64_t r949 = gl::sub(r935, r938);
  uint64_t r950 = gl::mul(r919, r930);
  uint64_t r951 = gl::mul(r913, r915);
  uint64_t r952 = gl::add(r934, r917);
  uint64_t r953 = gl::mul(r942, r920);
  uint64_t r954 = gl::sub(r946, r931);
  uint64_t r955 = gl::mul(r935, r934);
  uint64_t r956 = gl::mul(r942, r918);
  uint64_t r957 = gl::mul(r917, r953);
  uint64_t r958 = gl::add(r938, r918);
  uint64_t r959 = gl::add(r921, r924);
  uint64_t r960 = gl::mul(r923, r922);
  uint64_t r961 = gl::add(r944, r948);
  uint64_t r962 = gl::mul(r926, r929);
  uint64_t r963 = gl::mul(r924, r939);
  uint64_t r964 = gl::mul(r947, r949);
  uint64_t r965 = gl::add(r964, r944);
  uint64_t r966 = gl::add(r952, r926);
  uint64_t r967 = gl::add(r960, r956);
  uint64_t r968 = gl::mul(r930, r957);
  uint64_t r969 = gl::add(r934, r954);
  uint64_t r970 = gl::sub(r941, r963);
  uint64_t r971 = gl::mul(r932, r941);
  uint64_t r972 = gl::mul(r944, r965);
  uint64_t r973 = gl::mul(r946, r943);
  uint64_t r974 = gl::sub(r948, r968);
  uint64_t r975 = gl::add(r955, r935);
  uint64_t r976 = gl::sub(r943, r958);
  uint64_t r977 = gl::sub(r962, r956);
  uint64_t r978 = gl::add(r940, r951);
  uint64_t r979 = gl::add(r944, r960);
  uint64_t r980 = gl::mul(r978, r976);
  uint64_t r981 = gl::sub(r952, r965);
  uint64_t r982 = gl::add(r962, r967);
  uint64_t r983 = gl::add(r964, r947);
  uint64_t r984 = gl::mul(r946, r972);
  uint64_t r985 = gl::sub(r963, r951);
  uint64_t r986 = gl::mul(r948, r955);
  uint64_t r987 = gl::sub(r954, r962);
  uint64_t r988 = gl::sub(r968, r971);
  uint64_t r989 = gl::add(r970, r972);
  uint64_t r990 = gl::sub(r986, r979);
  uint64_t r991 = gl::mul(r977, r985);
  uint64_t r992 = gl::add(r979, r989);
  uint64_t r993 = gl::mul(r982, r992);
  uint64_t r994 = gl::sub(r975, r984);
  uint64_t r995 = gl::add(r969, r972);
  uint64_t r996 = gl::sub(r966, r975);
  uint64_t r997 = gl::mul(r985, r966);
  uint64_t r998 = gl::mul(r993, r967);
  uint64_t r999 = gl::mul(r991, r980);
  uint64_t r1000 = gl::mul(r970, r967);
  uint64_t r1001 = gl::sub(r983, r987);
  uint64_t r1002 = gl::sub(r993, r986);
  uint64_t r1003 = gl::mul(r977, r971);
  uint64_t r1004 = gl::add(r991, r972);
  uint64_t r1005 = gl::sub(r998, r980);
  uint64_t r1006 = gl::add(r966, r967);
  uint64_t r1007 = gl::mul(r1000, r1003);
  uint64_t r1008 = gl::add(r997, r968);
  uint64_t r1009 = gl::sub(r1005, r996);
  uint64_t r1010 = gl::sub(r983, r973);
  uint64_t r1011 = gl::sub(r982, r972);
  uint64_t r1012 = gl::mul(r980, r1007);
  uint64_t r1013 = gl::mul(r985, r1008);
  uint64_t r1014 = gl::add(r978, r1009);
  uint64_t r1015 = gl::add(r990, r1014);
  uint64_t acc = 0;
  acc = gl::add(acc, r952);
  acc = gl::add(acc, r953);
  acc = gl::add(acc, r954);
  acc = gl::add(acc, r955);
  acc = gl::add(acc, r956);
  acc = gl::add(acc, r957);
  acc = gl::add(acc, r958);
  acc = gl::add(acc, r959);
  acc = gl::add(acc, r960);
  acc = gl::add(acc, r961);
  acc = gl::add(acc, r962);
  acc = gl::add(acc, r963);
  acc = gl::add(acc, r964);
  acc = gl::add(acc, r965);
  acc = gl::add(acc, r966);
  acc = gl::add(acc, r967);
  acc = gl::add(acc, r968);
  acc = gl::add(acc, r969);
  acc = gl::add(acc, r970);
  acc = gl::add(acc, r971);
  acc = gl::add(acc, r972);
  acc = gl::add(acc, r973);
  acc = gl::add(acc, r974);
  acc = gl::add(acc, r975);
  acc = gl::add(acc, r976);
  acc = gl::add(acc, r977);
  acc = gl::add(acc, r978);
  acc = gl::add(acc, r979);
  acc = gl::add(acc, r980);
  acc = gl::add(acc, r981);
  acc = gl::add(acc, r982);
  acc = gl::add(acc, r983);
  acc = gl::add(acc, r984);
  acc = gl::add(acc, r985);
  acc = gl::add(acc, r986);
  acc = gl::add(acc, r987);
  acc = gl::add(acc, r988);
  acc = gl::add(acc, r989);
  acc = gl::add(acc, r990);
  acc = gl::add(acc, r991);
  acc = gl::add(acc, r992);
  acc = gl::add(acc, r993);
  acc = gl::add(acc, r994);
  acc = gl::add(acc, r995);
  acc = gl::add(acc, r996);
  acc = gl::add(acc, r997);
  acc = gl::add(acc, r998);
  acc = gl::add(acc, r999);
  acc = gl::add(acc, r1000);
  acc = gl::add(acc, r1001);
  acc = gl::add(acc, r1002);
  acc = gl::add(acc, r1003);
  acc = gl::add(acc, r1004);
  acc = gl::add(acc, r1005);
  acc = gl::add(acc, r1006);
  acc = gl::add(acc, r1007);
  acc = gl::add(acc, r1008);
  acc = gl::add(acc, r1009);
  acc = gl::add(acc, r1010);
  acc = gl::add(acc, r1011);
  acc = gl::add(acc, r1012);
  acc = gl::add(acc, r1013);
  acc = gl::add(acc, r1014);
  acc = gl::add(acc, r1015);
  out[blockIdx.x*128+threadIdx.x] = acc;
}
