// ntt_direct.hip — the "direct" NTT passes for gfx950 (interface: ntt_kernels.h).
//
// ntt_pass_wave_kernel moves every element through LDS four or five times per pass (staging exchange of the load, one
// exchange per radix round, gather exchange of the store) and synchronises its workgroup three times per tile. The passes
// here load a tile straight into the registers of the first radix round and store it straight from the registers of the
// last one; LDS is used only to change which index bits sit in registers:
//
//   column pass (R = 256 G rows, G = 1 / 2 / 4; tile = R rows x C = 64 / G adjacent columns, sixteen waves):
//     row m = (R/16) i + 16 g + w      i = register, g = lane / C, w = wave;  column c = lane % C  (8 bytes per lane, C lanes per
//     row segment: 128 bytes at R = 1024). Rounds: radix 16 over i (registers as loaded) -> twiddle w_R^(kA (16 g + w)) ->
//     exchange inside the wave (g <-> kA mod G) -> radix G over g -> twiddle w_16G^(kB w), wave-uniform -> exchange ACROSS
//     the waves (the one workgroup barrier pair of the tile) -> radix 16 over w -> inter-pass twiddle chain -> stores
//     from registers. Output frequency k1 = kA + 16 kB + 16 G kC.
//
// Two LDS round trips and two barriers per tile instead of four or five and three, no per-element address arithmetic
// (every LDS address is a per-lane base plus an immediate, every global address a per-lane offset plus a scalar base),
// and a workgroup keeps ONE column tile b for its whole life (it walks the polynomials and cosets of that tile), so the
// inter-pass twiddle chain of a lane is loop-invariant.
//
// Pipeline of a wave: the loads of tile k+1 are issued when the first rounds of tile k are done (the registers are free) and
// land while tile k is exchanged and finished; the sixteen (twiddle multiply, store) steps that end tile k are spread through
// the first rounds of tile k+1, so a wave issues a store every few dozen vector instructions instead of sixteen in a row,
// and no two of the workgroup's barrier-aligned waves queue at the memory pipeline with a burst. At the loop back-edge the
// tile's loads are the youngest vector-memory operations, which is what the compiler's counted vmcnt waits need to be exact.

#include "ntt_kernels.h"

namespace plonky2_hip {
namespace nttk {

namespace {

// Hook points of the diagnostic builds: tools/experiments/ntt_direct_diag.hip sets them and then includes this file (its results
// are wrong by design; tools/gpu_runs/ntt_direct_variants.sh). The product compiles this file as it stands: every hook is the
// identity / false.
#ifndef NTT_TAIL_GROUPED
#define NTT_TAIL_GROUPED 1
#endif
#ifndef DIRECT_DIAG_HOOKS
#define DIRECT_TILE_BARRIER() lds_barrier()
#define DIRECT_LOAD_TILE(t) (t)
#define DIRECT_STORE_TILE(t) (t)
#define DIRECT_DIAG_TAIL_FRONT_ON 0
#define DIRECT_DIAG_SAME_STORES_ON 0
#define DIRECT_NT_LOAD_COL false
#define DIRECT_NT_STORE_COL false
#define DIRECT_NT_LOAD_ROW false
#define DIRECT_NT_STORE_ROW false
#endif

__device__ __forceinline__ uint64_t lds_ld(const unsigned char *lds, uint32_t off) { return *reinterpret_cast<const uint64_t *>(lds + off); }
__device__ __forceinline__ void lds_st(unsigned char *lds, uint32_t off, uint64_t v) { *reinterpret_cast<uint64_t *>(lds + off) = v; }
template <bool NT = false>
__device__ __forceinline__ uint64_t g_ld(const uint64_t *base, uint32_t byte_off) {
    const uint64_t *q = reinterpret_cast<const uint64_t *>(reinterpret_cast<const unsigned char *>(base) + byte_off);
    if constexpr (NT) return __builtin_nontemporal_load(q);
    return *q;
}
template <bool NT = false>
__device__ __forceinline__ void g_st(uint64_t *base, uint32_t byte_off, uint64_t v) {
    uint64_t *q = reinterpret_cast<uint64_t *>(reinterpret_cast<unsigned char *>(base) + byte_off);
    if constexpr (NT)
        __builtin_nontemporal_store(v, q);
    else
        *q = v;
}
constexpr bool NT_LOAD_COL = DIRECT_NT_LOAD_COL, NT_STORE_COL = DIRECT_NT_STORE_COL, NT_LOAD_ROW = DIRECT_NT_LOAD_ROW, NT_STORE_ROW = DIRECT_NT_STORE_ROW;

template <int LOGG>
struct ColGeom {
    static constexpr int G = 1 << LOGG, LOGC = 6 - LOGG, C = 1 << LOGC, LOGR = 8 + LOGG, R = 1 << LOGR;
    static constexpr uint32_t ROWB = C * 8;                            // bytes of one (kAB, w) row of the exchange image
    // kAB stride: one pad row. For G = 4 (ROWB = 128) rows kAB and kAB + 1 then differ by 128 mod 256 bytes and every access of both
    // exchanges is conflict-free; for G = 8 (ROWB = 64) the stride is 1088 = 64 mod 256 and the first write of the private exchange keeps a
    // two-way conflict that only a row swizzle would remove (3 % of that kernel, profiles/r04_ntt_sizes.jsonl).
    static constexpr uint32_t SA = 16 * ROWB + (G >= 4 ? ROWB : 0);
    static constexpr uint32_t XBYTES = 16 * G * SA;
    static constexpr uint32_t TWBYTES = 16 * G * 16 * 8;               // per wave: w_R^(kA (16 g + w)), [g][kA]
    static constexpr uint32_t LDS_BYTES = XBYTES + TWBYTES;
    // coset LDE (first pass of coset_lde_batch): per coset z the powers of its shift that this pass needs, see the kernel
    static constexpr uint32_t cu_entries(uint32_t gz) { return gz * (G > 1 ? G * 16 : 256); }   // [z][g][i], or [z][w][i] when G = 1
    static constexpr uint32_t t2z_entries(uint32_t gz) { return G > 1 ? gz * 16 * G : 0; }         // [z][w][kB]
    static constexpr uint32_t cs_entries(uint32_t gz) { return gz * C; }                          // [z][c]
    static constexpr uint32_t coset_bytes(uint32_t gz) { return (cu_entries(gz) + t2z_entries(gz) + cs_entries(gz)) * 8; }
};

static_assert(ColGeom<3>::LDS_BYTES <= 160 * 1024 && ColGeom<2>::LDS_BYTES <= 160 * 1024, "the exchange image and the tables must fit the 160 KiB of LDS of a gfx950 CU");

// COSET (the first pass of the coset LDE, F_COSET): grid z = coset, every coset reads the same coefficients (in_sz = 0) scaled by the
// powers of its shift s_z and writes block bitrev(z). With j = m in_m + L (m = row = (R/16) i + 16 g + w, L = column):
//   s^(in_m ((R/16) i + 16 g))  multiplies the loaded element  (table CU [z][g][i] in LDS: depends on the register, so it cannot
//                               ride on a twiddle of an OUTPUT index),
//   s^(in_m w)                  is constant through the first two radix rounds and rides on their closing twiddle (T2Z [z][w][kB];
//                               for R = 256, which has no such twiddle, it is part of CU [z][w][i]),
//   s^L                         rides on the start of the inter-pass twiddle chain (CS [z][c]; the workgroup keeps its column tile).
// Sixteen + 16/G + 1 multiplications per lane and tile more than the plain pass; the tables are built once per workgroup.
// REVIN (round 6: the first pass of the two-pass INVERSE transform of 2^22 points, 2048 x 2048): the pass reads x'[j] = x[(n - j) mod n]
// instead of x[j] — ifft(x) = fft(x') / n (fft.rs:92-101 puts the same flip on the OUTPUT; on the input it is a load-address
// matter and the stores stay aligned). With j = m in_m + L (row m, column L, in_m = the number of columns):
//   L != 0:  x'[j] = x[(R - 1 - m) in_m + (in_m - L)]      L == 0:  x'[j] = x[((R - m) mod R) in_m]
// A lane's sixteen rows m = (R/16) i + 16 g + w are then sixteen addresses DEcreasing by ld_step; the one element that wraps
// (L = 0, m = 0 -> x[0]) is lane 0 of wave 0 of column tile 0, register 0. A tile's eight adjacent columns become a 64-byte
// segment read backwards and misaligned by one element, which the L2 absorbs on the load side (the natural-order forward
// plan's pass at 0.32 ms per 512 MiB runs 0.33 with it). 1 / n rides on the twiddle chain (chain_scale).
template <int LOGG, bool NATURAL, bool COSET = false, bool FINAL = false, bool REVIN = false>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(4, 4))) void ntt_col_direct_kernel(const PassParams p, const uint32_t gx, const uint32_t gy, const uint32_t gz, const uint32_t per_b) {
    using GEO = ColGeom<LOGG>;
    constexpr int G = GEO::G, LOGC = GEO::LOGC, C = GEO::C, LOGR = GEO::LOGR;
    constexpr uint32_t ROWB = GEO::ROWB, SA = GEO::SA;
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
    unsigned char *X = ldsb;
    unsigned char *TW = ldsb + GEO::XBYTES;
    unsigned char *CU = TW + GEO::TWBYTES, *T2Z = CU + GEO::cu_entries(gz) * 8, *CS = T2Z + GEO::t2z_entries(gz) * 8;   // COSET only

    const uint32_t tid = threadIdx.x, lane = tid & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t g = lane >> LOGC, c = lane & (C - 1);

    // ---- which tiles this workgroup walks -------------------------------------------------------------------------
    const uint32_t W = gridDim.x;
    const uint32_t u = (W & 7) == 0 ? (blockIdx.x & 7) * (W >> 3) + (blockIdx.x >> 3) : blockIdx.x;  // workgroups of one XCD take adjacent column tiles
    const uint32_t P = gy * gz;
    uint32_t b0, bstep, p0, pstep;
    if (per_b) {
        // COSET: the per_b workgroups of a column tile are neighbours (one XCD, one L2) and a workgroup takes the cosets of a
        // polynomial one after the other: all of them read the same coefficients
        if constexpr (COSET) b0 = u / per_b, p0 = u % per_b;
        else b0 = u % gx, p0 = u / gx;
        bstep = gx, pstep = per_b;
    } else {
        b0 = u, bstep = W, p0 = 0, pstep = 1;
    }
    const uint32_t np_local = p0 < P ? (P - p0 + pstep - 1) / pstep : 0;
    const uint32_t nb_local = b0 < gx ? (gx - b0 + bstep - 1) / bstep : 0;
    const uint32_t n_tiles = np_local * nb_local;
    if (n_tiles == 0) return;

    // ---- tables ------------------------------------------------------------------------------------------------------
    for (uint32_t e = tid; e < 16u * G * 16u; e += 1024) {
        const uint32_t w_ = e / (16 * G), g_ = (e / 16) % G, ka = e % 16;
        reinterpret_cast<uint64_t *>(TW)[e] = p.twh[(ka * (16 * g_ + w_)) << (12 - LOGR)];
    }
    uint64_t t2[G > 1 ? G : 1];  // w_16G^(kB w), wave-uniform
    if constexpr (!COSET)
        static_for<1, G>([&](auto K_) {
            constexpr int kb = decltype(K_)::value;
            t2[kb] = p.twh[(kb * wave) << (12 - 4 - LOGG)];
        });
    if constexpr (COSET) {
        auto cs_pow = [&](uint32_t z, uint64_t ex) { return gl::mul(p.cs_hi[z * p.cs_hi_len + (uint32_t)(ex >> 10)], p.cs_lo[z * 1024 + (uint32_t)(ex & 1023)]); };
        for (uint32_t e = tid; e < GEO::cu_entries(gz); e += 1024) {
            const uint32_t i = e & 15, mid = G > 1 ? 16 * ((e >> 4) % G) : (e >> 4) & 15, z = G > 1 ? e / (16 * G) : e >> 8;
            reinterpret_cast<uint64_t *>(CU)[e] = cs_pow(z, (uint64_t)p.in_m * (((uint32_t)i << (LOGR - 4)) + mid));
        }
        for (uint32_t e = tid; e < GEO::t2z_entries(gz); e += 1024) {
            const uint32_t kb = e % G, w_ = (e / G) & 15, z = e / (16 * G);
            reinterpret_cast<uint64_t *>(T2Z)[e] = gl::mul(p.twh[(kb * w_) << (12 - 4 - LOGG)], cs_pow(z, (uint64_t)p.in_m * w_));
        }
        for (uint32_t e = tid; e < GEO::cs_entries(gz); e += 1024)   // the launcher makes sure the workgroup keeps one column tile (per_b != 0)
            reinterpret_cast<uint64_t *>(CS)[e] = cs_pow(e / C, (uint64_t)b0 * C + (e % C));
    }

    // ---- per-lane constants ------------------------------------------------------------------------------------------
    // Everything below is a function of (lane, wave) and the same for every tile. What the first rounds use stays in registers;
    // what is used once per tile (global offsets, the address of the exchange read) is recomputed from an opaque copy of the
    // lane index where it is needed, so that it is not kept live through the register-hungry part of the loop.
    const uint32_t tw_base = (wave * G + g) * 128;
    const uint32_t lane_x = wave * ROWB + c * 8;               // this wave's slots of the exchange image: [row][wave][c]
    const uint32_t pw_base = 16 * g * SA + lane_x;             // private exchange, written as row 16 g + kA
    const uint32_t pr_base = g * SA + lane_x;                  // ... read back as row 16 g'' + (g + G j); also where round 2 leaves kAB = g + G j + 16 kB
    constexpr bool natural = NATURAL;
    auto opaque_lane = [&]() {
        uint32_t l = lane;
        asm volatile("" : "+v"(l));
        return l;
    };
    // the output block a lane finishes: rows kAB + 16 G kC, kAB = G wave + g
    auto kab_of = [&](uint32_t l) { return G * wave + (l >> LOGC); };
    auto xr_base_of = [&](uint32_t l) { return kab_of(l) * SA + (l & (C - 1)) * 8; };   // + w'' * ROWB
    auto ld_off_of = [&](uint32_t l) { return (uint32_t)(((l & (C - 1)) + (uint64_t)(16 * (l >> LOGC) + wave) * p.in_m) * 8); };
    // row of frequency k1 = kAB + 16 G kC in the output: k1 itself, or its bit reversal when the transform runs in place
    const uint64_t out_c = p.out_c ? p.out_c : 1;   // stride between the tile's columns in the output (N1 with out_m = 1: transposed store)
    auto st_off_of = [&](uint32_t l) {
        const uint32_t kab = kab_of(l);
        const uint32_t out_row_lane = natural ? kab : (brev_rt(kab & 15, 4) << (LOGR - 4)) | (brev_rt(kab >> 4, LOGG) << 4);
        return (uint32_t)(((l & (C - 1)) * out_c + (uint64_t)out_row_lane * p.out_m) * 8);
    };

    auto tile_of = [&](uint32_t t, uint32_t &b, uint32_t &a, uint32_t &z) {
        const uint32_t bi = t / np_local, pi = t - bi * np_local;
        b = b0 + bi * bstep;
        const uint32_t pp = p0 + pi * pstep;
        if constexpr (COSET) a = pp / gz, z = pp % gz;
        else a = pp % gy, z = pp / gy;
    };

    uint64_t A[16];   // tile in flight / first rounds
    const uint32_t ld_step = (uint32_t)(((uint64_t)p.in_m << (LOGR - 4)) * 8);   // register i holds row (R/16) i + ...
    auto issue_loads = [&](uint32_t t) {
        uint32_t b, a, z;
        tile_of(DIRECT_LOAD_TILE(t), b, a, z);
        if constexpr (REVIN) {
            static_assert(!COSET && NATURAL, "the reversed-input pass is the first pass of a natural-order inverse transform");
            const uint64_t *base = p.src + (a * p.in_sa + z * p.in_sz);
            const uint32_t l = opaque_lane();
            const uint32_t L = b * C + (l & (C - 1)), m0 = 16 * (l >> LOGC) + wave;
            const uint32_t cols = (uint32_t)p.in_m;
            // byte offset of register 0's element; the wrapping element keeps the VIRTUAL row R (one past the end) for the steps below
            uint32_t off = L ? (((uint32_t)GEO::R - 1 - m0) * cols + (cols - L)) * 8 : (((uint32_t)GEO::R - m0) * cols) * 8;
            const bool wraps = L == 0 && m0 == 0;
            static_for<0, 16>([&](auto I_) {
                constexpr int i = decltype(I_)::value;
                if constexpr (i == 0) A[i] = g_ld<NT_LOAD_COL>(base, wraps ? 0u : off);
                else A[i] = g_ld<NT_LOAD_COL>(base, off);
                if constexpr (i < 15) off -= ld_step;
            });
            return;
        }
        const uint64_t *base = p.src + (a * p.in_sa + b * p.in_sb + z * p.in_sz);
        uint32_t off = ld_off_of(opaque_lane());  // tile-invariant, like the sixteen offsets derived from it: left to itself the compiler keeps them all in registers
        static_for<0, 16>([&](auto I_) {
            constexpr int i = decltype(I_)::value;
            A[i] = g_ld<NT_LOAD_COL>(base, off);
            if constexpr (i < 15) off += ld_step;
        });
    };

    uint64_t B[16];   // tile being finished: sixteen w for one (kAB, c)
    uint64_t cc0 = 1, cstep = 1;   // inter-pass twiddle chain of this lane: w^(L kAB) (times 1/n, coset power), w^(L 16 G)
    // Without cosets a lane's sixteen twiddles are the same for every tile of the workgroup (it keeps its column tile): the first
    // eight stay in registers, the other eight are one multiplication by step^8 away — 24 multiplications per tile instead of the
    // chain's 31. (With cosets the chain starts from a different power per tile and is walked as before.)
    constexpr bool kept_twiddles = !COSET && !FINAL;
    uint64_t cw[kept_twiddles ? 8 : 1], cstep8 = 1;
    uint32_t chain_b = 0xFFFFFFFFu;
    const uint32_t st_row = (uint32_t)(p.out_m * 8);

    // One step of the end of tile `t`: output kC = j gets its inter-pass twiddle and is stored.
    uint64_t cc = 1;
    uint64_t *obase = p.dst;
    uint32_t so = 0;
    auto tail_begin = [&](uint32_t k) {
        uint32_t b, a, z;
        tile_of(k, b, a, z);
        if (!FINAL && b != chain_b) {
            chain_b = b;
            const uint32_t l = opaque_lane();
            const uint64_t L = (uint64_t)b * C + (l & (C - 1));
            cc0 = wpow(p, L * kab_of(l));
            if (p.chain_scale != 1) cc0 = gl::mul(cc0, p.chain_scale);
            cstep = wpow(p, L << (4 + LOGG));
            // the look-ups are complete when this block ends: at the join below the compiler would otherwise wait for "possibly
            // pending" loads with vmcnt(0) in every iteration, which also waits for the prefetched tile
            asm volatile("" : "+v"(cc0), "+v"(cstep));
            if constexpr (kept_twiddles) {
                cw[0] = cc0;
                static_for<1, 8>([&](auto J_) {
                    constexpr int j = decltype(J_)::value;
                    cw[j] = gl::mul(cw[j - 1], cstep);
                });
                const uint64_t s2 = gl::mul(cstep, cstep), s4 = gl::mul(s2, s2);
                cstep8 = gl::mul(s4, s4);
            }
        }
        if constexpr (DIRECT_DIAG_SAME_STORES_ON) tile_of(0, b, a, z);
        obase = p.dst + (a * p.out_sa + b * p.out_sb + (COSET ? brev_rt(z, p.rate_bits) : z) * p.out_sz);
        if constexpr (COSET) cc = gl::mul(cc0, lds_ld(CS, (z * C + (opaque_lane() & (C - 1))) * 8));
        else cc = cc0;
        so = st_off_of(opaque_lane());
    };
    auto tail_unit = [&](auto J_) {
        constexpr int j = decltype(J_)::value;   // kC
        constexpr int s3 = brev_c(j, 4);
        if constexpr (FINAL) {
            B[s3] = gl::canon(B[s3]);   // the last pass of a transform: no inter-pass twiddle, a boundary buffer gets canonical values
        } else if constexpr (kept_twiddles) {
            if constexpr (j < 8) B[s3] = gl::mul(B[s3], cw[j]);
            else B[s3] = gl::mul(B[s3], gl::mul(cw[j - 8], cstep8));
        } else {
            B[s3] = gl::mul(B[s3], cc);
            if constexpr (j < 15) cc = gl::mul(cc, cstep);
        }
        const uint32_t row = natural ? (uint32_t)(j << (4 + LOGG)) : (uint32_t)s3;
        g_st<NT_STORE_COL>(obase, so + row * st_row, B[s3]);   // unconditional: a branch here would make the compiler forget how many stores are pending
        __builtin_amdgcn_sched_barrier(0);
    };
    // the store of tail step j alone (its twiddle already applied)
    auto tail_store = [&](auto J_) {
        constexpr int j = decltype(J_)::value;
        constexpr int s3 = brev_c(j, 4);
        const uint32_t row = natural ? (uint32_t)(j << (4 + LOGG)) : (uint32_t)s3;
        g_st<NT_STORE_COL>(obase, so + row * st_row, B[s3]);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto tail_units = [&](auto LO_, auto HI_) {
        constexpr int LO = decltype(LO_)::value, HI = decltype(HI_)::value, N = HI - LO;
        if constexpr (NTT_TAIL_GROUPED && kept_twiddles && N >= 1 && N <= 8) {
            // the multiplications of the steps [LO, HI) as one group with deferred rare paths (gl_field.h): first the twiddles that are
            // one multiplication away (steps 8-15: cw[j - 8] step^8), then the outputs; then the stores in order
            uint64_t tw[N];
            constexpr int N_HI = HI > 8 ? HI - (LO > 8 ? LO : 8) : 0;   // steps >= 8 in the range
            if constexpr (N_HI > 0)
                rare_group<N_HI>([&](auto I_, gl::rare_mask &f) { constexpr int j = (LO > 8 ? LO : 8) + decltype(I_)::value; tw[j - LO] = gl::mul_f(cw[j - 8], cstep8, f); },
                                 [&](auto I_, gl::rare_mask f) { constexpr int j = (LO > 8 ? LO : 8) + decltype(I_)::value; tw[j - LO] = gl::mul_fix(tw[j - LO], f); });
            rare_group<N>(
                [&](auto I_, gl::rare_mask &f) {
                    constexpr int j = LO + decltype(I_)::value, s3 = brev_c(j, 4);
                    if constexpr (j < 8) B[s3] = gl::mul_f(B[s3], cw[j], f);
                    else B[s3] = gl::mul_f(B[s3], tw[j - LO], f);
                },
                [&](auto I_, gl::rare_mask f) {
                    constexpr int s3 = brev_c(LO + decltype(I_)::value, 4);
                    B[s3] = gl::mul_fix(B[s3], f);
                });
            static_for<LO, HI>([&](auto J_) { tail_store(J_); });
        } else {
            static_for<LO, HI>([&](auto J_) { tail_unit(J_); });
        }
    };

    // first rounds of the tile whose elements are in A: radix 16 over i, twiddle, exchange inside the wave, radix G over g,
    // twiddle, and the results into the exchange image; the sixteen tail steps of the previous tile are spread through it
    auto first_rounds = [&](auto WITH_TAIL_, uint32_t t_in_a) {
        constexpr bool with_tail = decltype(WITH_TAIL_)::value;   // false: the prologue, nothing to finish
        uint32_t zc = 0;   // COSET: the coset of the tile whose elements are in A
        if constexpr (COSET) {
            uint32_t b_, a_;
            tile_of(t_in_a, b_, a_, zc);
        }
        if constexpr (with_tail && DIRECT_DIAG_TAIL_FRONT_ON) tail_units(std::integral_constant<int, 0>{}, std::integral_constant<int, 16>{});
#define TAIL(lo, hi) do { if constexpr (with_tail && !DIRECT_DIAG_TAIL_FRONT_ON) tail_units(std::integral_constant<int, lo>{}, std::integral_constant<int, hi>{}); } while (0)
        TAIL(0, 4);   // four results leave before the butterflies need their temporaries
        if constexpr (COSET) {
            const uint32_t cu_base = (G > 1 ? (zc * G + (opaque_lane() >> LOGC)) * 16 : (zc * 16 + wave) * 16) * 8;
            mul_run<0, 16>(A, [](auto I_) { return decltype(I_)::value; }, [&](auto I_) { return lds_ld(CU, cu_base + decltype(I_)::value * 8); }, [](auto) {});
        }
        radix_dif_stage<4, 0, 3>(A);
        TAIL(4, 6);
        radix_dif_stage<4, 0, 2>(A);
        TAIL(6, 8);
        radix_dif_stage<4, 0, 1>(A);
        TAIL(8, 10);
        radix_dif_stage<4, 0, 0>(A);
        TAIL(10, 12);
        mul_run<1, 16>(A, [](auto S_) { return decltype(S_)::value; }, [&](auto S_) { return lds_ld(TW, tw_base + brev_c(decltype(S_)::value, 4) * 8); },
                       [&](auto S_) {
                           constexpr int s = decltype(S_)::value;
                           if constexpr (s == 5) TAIL(12, 13);
                           if constexpr (s == 10) TAIL(13, 14);
                       });
        if constexpr (G > 1) {
            static_for<0, 16>([&](auto S_) {
                constexpr int s = decltype(S_)::value;
                constexpr int ka = brev_c(s, 4);
                lds_st(X, pw_base + ka * SA, A[s]);
            });
            tile_sync<64>();
            static_for<0, 16 / G>([&](auto J_) {
                constexpr int j = decltype(J_)::value;
                static_for<0, G>([&](auto GG_) {
                    constexpr int gg = decltype(GG_)::value;
                    A[j * G + gg] = lds_ld(X, pr_base + (16 * gg + G * j) * SA);
                });
            });
            tile_sync<64>();
            radix_dif_blocks<LOGG>(A);   // the 16 / G radix-G butterflies stage by stage together: larger groups for the deferred rare paths
            TAIL(14, 15);
            // register e = j G + s2 takes w_16G^(kB w), kB = bitrev(s2); without cosets the factor of kB = 0 is 1 and skipped
            if constexpr (COSET)
                mul_run<0, 16>(A, [](auto E_) { return decltype(E_)::value; },
                               [&](auto E_) { return lds_ld(T2Z, ((zc * 16 + wave) * G + brev_c(decltype(E_)::value % G, LOGG)) * 8); }, [](auto) {});
            else  // the (G - 1) 16 / G registers with kB != 0: k -> j = k / (G - 1), s2 = 1 + k % (G - 1)
                mul_run<0, (G - 1) * (16 / G)>(A, [](auto K_) { return (decltype(K_)::value / (G - 1)) * G + 1 + decltype(K_)::value % (G - 1); },
                                               [&](auto K_) { return t2[brev_c(1 + decltype(K_)::value % (G - 1), LOGG)]; }, [](auto) {});
            static_for<0, 16 / G>([&](auto J_) {
                constexpr int j = decltype(J_)::value;
                static_for<0, G>([&](auto S_) {
                    constexpr int s2 = decltype(S_)::value;
                    constexpr int kb = brev_c(s2, LOGG);
                    lds_st(X, pr_base + (G * j + 16 * kb) * SA, A[j * G + s2]);
                });
            });
            TAIL(15, 16);
        } else {
            TAIL(14, 16);
            static_for<0, 16>([&](auto S_) {
                constexpr int s = decltype(S_)::value;
                constexpr int ka = brev_c(s, 4);
                lds_st(X, pr_base + ka * SA, A[s]);
            });
        }
#undef TAIL
    };

    // ---- pipeline ---------------------------------------------------------------------------------------------------
    // Iteration k: the exchange image of tile k is read (two barriers), its radix 16 over w is done, and its tail runs inside the
    // first rounds of tile k+1.
    lds_barrier();  // tables
    issue_loads(0);
    const uint32_t last = n_tiles - 1;
    first_rounds(std::false_type{}, 0);
    if (last > 0) issue_loads(1);
#pragma unroll 1
    for (uint32_t k = 0; k <= last; k++) {
        DIRECT_TILE_BARRIER();  // image of tile k complete
        const uint32_t xr_base = xr_base_of(opaque_lane());
        static_for<0, 16>([&](auto WR_) {
            constexpr int wr = decltype(WR_)::value;
            B[wr] = lds_ld(X, xr_base + wr * ROWB);
        });
        DIRECT_TILE_BARRIER();  // everyone has read it: the slots may be rewritten
        radix_dif<4, 0>(B);
        tail_begin(k);
        if (k < last) {
            first_rounds(std::true_type{}, k + 1);
            if (k + 1 < last) issue_loads(k + 2);
        } else {
            tail_units(std::integral_constant<int, 0>{}, std::integral_constant<int, 16>{});
        }
    }
}

template <int LOGG, bool NATURAL, bool COSET = false, bool FINAL = false, bool REVIN = false>
hipError_t launch_col_direct_t(const PassParams &p, dim3 grid, hipStream_t stream) {
    using GEO = ColGeom<LOGG>;
    const uint32_t lds_bytes = GEO::LDS_BYTES + (COSET ? GEO::coset_bytes(grid.z) : 0);
    static DynamicLds attr;
    if (hipError_t e = allow_dynamic_lds(attr, reinterpret_cast<const void *>(&ntt_col_direct_kernel<LOGG, NATURAL, COSET, FINAL, REVIN>), lds_bytes); e != hipSuccess) return e;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const uint64_t pairs = (uint64_t)grid.y * grid.z, total = pairs * grid.x;
    if (total == 0) return hipSuccess;
    if (total > 0xFFFFFFFFull) return hipErrorInvalidValue;
    // one workgroup per CU (the exchange image is 128 KiB + pad); a workgroup keeps one column tile when there are no more
    // column tiles than CUs, and then the polynomials and cosets are dealt to the cus / grid.x workgroups of that tile
    uint32_t per_b = 0, wgs;
    if (grid.x <= (uint32_t)cus) {
        per_b = (uint32_t)cus / grid.x;
        if (per_b > pairs) per_b = (uint32_t)pairs;
        wgs = grid.x * per_b;
    } else {
        wgs = (uint32_t)cus;
    }
    if (COSET && per_b == 0) return hipErrorInvalidValue;   // col_direct_coset_ok() said otherwise
    hipLaunchKernelGGL((ntt_col_direct_kernel<LOGG, NATURAL, COSET, FINAL, REVIN>), dim3(wgs), dim3(1024), lds_bytes, stream, p, (uint32_t)grid.x, (uint32_t)grid.y, (uint32_t)grid.z, per_b);
    return hipGetLastError();
}


// ---------------------------------------------------------------------------------------------
// Row pass with natural-order (transposed) output, rows of R = 1024 points; tile = 16 adjacent rows, one per wave.
//   position j2 = 64 i + 4 h + q     i = register, (h, q) = lane: 8 bytes per lane, 512 contiguous bytes per load.
//   radix 16 over i -> twiddle w_R^(kA lane) -> exchange inside the wave (h <-> kA) -> radix 16 over h -> exchange ACROSS the
//   waves: wave (kBhi, kAlo), lane (kAhi, row), registers (kBlo, q) -> twiddle w_64^(kB q) AS SHIFTS (every 64th root of unity is a
//   power of two; kBlo and q are register indices and kBhi is wave-uniform, so the shift amounts are compile-time constants behind
//   a four-way scalar switch: ntt_kernels.h, shift_twiddles_radix4) -> radix 4 over q ->
//   stores of X[k1 + N1 k2], k2 = kA + 16 kB + 256 kC: sixteen lanes = sixteen adjacent rows k1 = one 128-byte segment.
// The exchange image is [kA][kB][q][row] with pads chosen so that every access of both exchanges is conflict-free and the
// slots a wave writes (its row's column of the image) are also the slots of its private exchange:
//   kA stride 8736, kB stride 544, q stride 136, row stride 8 bytes.
// ---------------------------------------------------------------------------------------------
struct RowGeom {
    static constexpr uint32_t SQ = 136, SB = 544, SA = 8736;
    static constexpr uint32_t XBYTES = 16 * SA;
    static constexpr uint32_t TW1_STRIDE = 136, TW1_BYTES = 64 * TW1_STRIDE;   // [lane][kA]: w_1024^(kA lane)
    static constexpr uint32_t TW2_STRIDE = 136, TW2_BYTES = 4 * TW2_STRIDE;    // [q][kB]:   w_64^(kB q), in-place pass only
    static constexpr uint32_t LDS_BYTES = XBYTES + TW1_BYTES;   // the natural-order pass has no table for the second twiddle (shifts)
};

template <bool INVERSE>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(4, 4))) void ntt_row_natural_direct_kernel(const PassParams p, const uint32_t gx, const uint32_t gy, const uint32_t gz) {
    using GEO = RowGeom;
    constexpr uint32_t SQ = GEO::SQ, SB = GEO::SB, SA = GEO::SA;
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
    unsigned char *X = ldsb;
    unsigned char *TW1 = ldsb + GEO::XBYTES;

    const uint32_t tid = threadIdx.x, lane = tid & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    const uint32_t W = gridDim.x;
    const uint32_t u = (W & 7) == 0 ? (blockIdx.x & 7) * (W >> 3) + (blockIdx.x >> 3) : blockIdx.x;  // workgroups of one XCD take adjacent row tiles
    const uint32_t total = gx * gy * gz;
    if (u >= total) return;
    const uint32_t n_tiles = (total - u + W - 1) / W;

    for (uint32_t e = tid; e < 64 * 16; e += 1024) {
        const uint32_t l = e >> 4, ka = e & 15;
        *reinterpret_cast<uint64_t *>(TW1 + l * GEO::TW1_STRIDE + ka * 8) = p.twh[(ka * l) << 2];
    }

    // first rounds: lane = (h, q) before the private exchange, (kA', q) after it
    const uint32_t q = lane & 3, hi4 = lane >> 2;
    const uint32_t tw1_base = lane * GEO::TW1_STRIDE;
    const uint32_t pw_base = hi4 * SB + q * SQ + wave * 8;      // + kA * SA
    const uint32_t pr_base = hi4 * SA + q * SQ + wave * 8;      // + h * SB; round 2 leaves (kA', kB, q) at + kB * SB
    auto opaque_lane = [&]() {
        uint32_t l = lane;
        asm volatile("" : "+v"(l));
        return l;
    };

    auto tile_of = [&](uint32_t t, uint32_t &b, uint32_t &a, uint32_t &z) {
        const uint32_t id = u + t * W;
        b = id % gx;
        const uint32_t r = id / gx;
        a = r % gy;
        z = r / gy;
    };

    uint64_t A[16];
    auto issue_loads = [&](uint32_t t) {
        uint32_t b, a, z;
        tile_of(DIRECT_LOAD_TILE(t), b, a, z);
        const uint32_t row = (b * 16 + wave + p.row_shift) & (p.t_limit - 1);   // inverse: the tile is rotated by one row (see ntt.hip)
        const uint64_t *base = p.src + (a * p.in_sa + z * p.in_sz + (uint64_t)row * p.in_t);
        const uint32_t off = opaque_lane() * 8;
        static_for<0, 16>([&](auto I_) {
            constexpr int i = decltype(I_)::value;
            A[i] = g_ld<NT_LOAD_ROW>(base, off + i * 512);
        });
    };

    uint64_t B[16];   // registers (kBlo, q); wave = (kBhi, kAlo), lane = (kAhi, row)
    const uint32_t kbhi = wave >> 2, kalo = wave & 3;
    uint64_t *obase = p.dst;
    uint32_t o_lane = 0;   // element index inside the polynomial of this lane's outputs, before the register part
    const uint32_t n_mask = (1u << p.log_n) - 1;
    auto tail_begin = [&](uint32_t k) {
        uint32_t b, a, z;
        tile_of(DIRECT_STORE_TILE(k), b, a, z);
        obase = p.dst + (a * p.out_sa + z * p.out_sz);
        const uint32_t l = opaque_lane();
        const uint32_t r = l & 15, ka = 4 * (l >> 4) + kalo;
        const uint32_t k1 = (b * 16 + r + p.row_shift) & (p.t_limit - 1);
        o_lane = k1 + (uint32_t)p.out_m * (ka + 64 * kbhi);
    };
    // register slot s of B: kBlo = s >> 2, and after the radix 4 over q slot (s & 3) holds kC = bitrev2(s & 3)
    auto tail_unit = [&](auto J_) {
        constexpr int s = decltype(J_)::value;
        constexpr int kblo = s >> 2, kc = brev_c(s & 3, 2);
        uint32_t o = o_lane + (uint32_t)p.out_m * (16 * kblo + 256 * kc);
        if constexpr (INVERSE) o = (0u - o) & n_mask;   // index flip i -> n - i of the inverse transform (fft.rs:92-101)
        g_st<NT_STORE_ROW>(obase, o * 8, gl::canon(B[s]));
        __builtin_amdgcn_sched_barrier(0);
    };
    auto tail_units = [&](auto LO_, auto HI_) {
        static_for<decltype(LO_)::value, decltype(HI_)::value>([&](auto J_) { tail_unit(J_); });
    };

    auto first_rounds = [&](auto WITH_TAIL_) {
        constexpr bool with_tail = decltype(WITH_TAIL_)::value;
        if constexpr (with_tail && DIRECT_DIAG_TAIL_FRONT_ON) tail_units(std::integral_constant<int, 0>{}, std::integral_constant<int, 16>{});
#define TAIL(lo, hi) do { if constexpr (with_tail && !DIRECT_DIAG_TAIL_FRONT_ON) tail_units(std::integral_constant<int, lo>{}, std::integral_constant<int, hi>{}); } while (0)
        TAIL(0, 3);
        radix_dif_stage<4, 0, 3>(A);
        TAIL(3, 5);
        radix_dif_stage<4, 0, 2>(A);
        TAIL(5, 6);
        radix_dif_stage<4, 0, 1>(A);
        TAIL(6, 7);
        radix_dif_stage<4, 0, 0>(A);
        TAIL(7, 8);
        mul_run<1, 16>(A, [](auto S_) { return decltype(S_)::value; }, [&](auto S_) { return lds_ld(TW1, tw1_base + brev_c(decltype(S_)::value, 4) * 8); },
                       [&](auto S_) {
                           if constexpr (decltype(S_)::value == 8) TAIL(8, 9);
                       });
        static_for<0, 16>([&](auto S_) {
            constexpr int s = decltype(S_)::value;
            constexpr int ka = brev_c(s, 4);
            lds_st(X, pw_base + ka * SA, A[s]);
        });
        tile_sync<64>();
        static_for<0, 16>([&](auto H_) {
            constexpr int h = decltype(H_)::value;
            A[h] = lds_ld(X, pr_base + h * SB);
        });
        tile_sync<64>();
        TAIL(9, 10);
        radix_dif_stage<4, 0, 3>(A);
        TAIL(10, 11);
        radix_dif_stage<4, 0, 2>(A);
        TAIL(11, 12);
        radix_dif_stage<4, 0, 1>(A);
        TAIL(12, 13);
        radix_dif_stage<4, 0, 0>(A);
        TAIL(13, 14);
        static_for<0, 16>([&](auto S_) {
            constexpr int s = decltype(S_)::value;
            constexpr int kb = brev_c(s, 4);
            lds_st(X, pr_base + kb * SB, A[s]);
            if constexpr (s == 7) TAIL(14, 15);
        });
        TAIL(15, 16);
#undef TAIL
    };

    lds_barrier();  // tables
    issue_loads(0);
    const uint32_t last = n_tiles - 1;
    first_rounds(std::false_type{});
    if (last > 0) issue_loads(1);
#pragma unroll 1
    for (uint32_t k = 0; k <= last; k++) {
        DIRECT_TILE_BARRIER();  // image of tile k complete
        {
            const uint32_t l = opaque_lane();
            // conflict-free like the writes: the sixteen rows are 128 contiguous bytes, kAhi moves by 4 SA = 128 (mod 256) bytes
            const uint32_t xr_base = (4 * (l >> 4) + kalo) * SA + kbhi * (4 * SB) + (l & 15) * 8;
            static_for<0, 16>([&](auto S_) {
                constexpr int s = decltype(S_)::value;
                B[s] = lds_ld(X, xr_base + (s >> 2) * SB + (s & 3) * SQ);
            });
        }
        DIRECT_TILE_BARRIER();  // everyone has read it
        // twiddle w_64^(kB q) and the radix 4 over q; kbhi is wave-uniform: a scalar switch
        if (kbhi == 0) shift_twiddles_radix4<0>(B);
        else if (kbhi == 1) shift_twiddles_radix4<1>(B);
        else if (kbhi == 2) shift_twiddles_radix4<2>(B);
        else shift_twiddles_radix4<3>(B);
        tail_begin(k);
        if (k < last) {
            first_rounds(std::true_type{});
            if (k + 1 < last) issue_loads(k + 2);
        } else {
            tail_units(std::integral_constant<int, 0>{}, std::integral_constant<int, 16>{});
        }
    }
}

template <bool INVERSE>
hipError_t launch_row_natural_direct_t(const PassParams &p, dim3 grid, hipStream_t stream) {
    static DynamicLds attr;
    if (hipError_t e = allow_dynamic_lds(attr, reinterpret_cast<const void *>(&ntt_row_natural_direct_kernel<INVERSE>), RowGeom::LDS_BYTES); e != hipSuccess) return e;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const uint64_t total = (uint64_t)grid.x * grid.y * grid.z;
    if (total == 0) return hipSuccess;
    if (total > 0xFFFFFFFFull) return hipErrorInvalidValue;
    const uint32_t wgs = (uint32_t)(total < (uint64_t)cus ? total : (uint64_t)cus);
    hipLaunchKernelGGL((ntt_row_natural_direct_kernel<INVERSE>), dim3(wgs), dim3(1024), RowGeom::LDS_BYTES, stream, p, (uint32_t)grid.x, (uint32_t)grid.y, (uint32_t)grid.z);
    return hipGetLastError();
}


// ---------------------------------------------------------------------------------------------
// Row pass in place (bit-reversed output: every pass of the LDE / commit path's last stage), rows of R = 1024 points, one row
// per wave, no workgroup barrier at all: three exchanges inside the wave through its own 1084-slot buffer.
//   load as the natural-order row pass (position 64 i + lane) -> radix 16 over i -> twiddle -> E1 (h <-> kA) -> radix 16 over h
//   -> twiddle w_64^(kB q) -> E2 (lane (kA, kBhi), registers (kBlo, q)) -> radix 4 over q -> E3 into the store layout
//   (register = position >> 6, lane = position & 63, position = bitrev10(k2)) -> 512 contiguous bytes per store.
// Slots (8 bytes each; every access of the three exchanges is conflict-free, tests/ntt_direct_inplace_model.py):
//   E1: 4 kA + 68 h + q        E2: 64 kB + 4 kA + (q ^ (kB >> 2))        E3: (pos & ~3) | ((pos & 3) ^ (pos >> 8))
// Pipeline: [E1 read .. stores of row k] then [first round of row k+1 -> E1 write] then the loads of row k+2: at the loop
// back-edge the loads are the youngest vector-memory operations (see the column pass).
// ---------------------------------------------------------------------------------------------
struct RowInplaceGeom {
    static constexpr uint32_t WBUF_BYTES = 1084 * 8;   // per wave
    static constexpr uint32_t XBYTES = 16 * WBUF_BYTES;
    static constexpr uint32_t LDS_BYTES = XBYTES + RowGeom::TW1_BYTES + RowGeom::TW2_BYTES;
    // HALVES (rows of 2048 points): [lane][i] = w_2048^(lane + 64 i), the twiddles of the radix-2 stage that is formed on load
    static constexpr uint32_t TW0_STRIDE = 136, TW0_BYTES = 64 * TW0_STRIDE;
    static constexpr uint32_t LDS_BYTES_HALVES = LDS_BYTES + TW0_BYTES;
};
static_assert(RowInplaceGeom::LDS_BYTES_HALVES <= 160 * 1024, "the wave buffers and the three twiddle tables must fit the 160 KiB of LDS");

// HALVES (round 6): rows of 2048 points, two waves per row. A 2048-point DIF transform is one radix-2 stage — x_j + x_(j+1024) feeds the
// even frequencies, (x_j - x_(j+1024)) w_2048^j the odd ones — followed by two independent 1024-point transforms whose bit-reversed
// outputs are the two halves of the row: frequency k = 2 k' + h sits at position h 1024 + bitrev10(k'). Unit u = (row, h): the wave
// loads BOTH halves of the row (its partner wave, the next unit, reads the same 16 KiB a moment later: L2), forms its half's input of
// the 1024-point pipeline below on the way in, and stores into its half. One general multiplication per element of the odd half
// (table TW0); the even half costs an addition. NOT in place: nothing orders one wave's stores behind its partner's loads of the same
// row, so the planner gives this form a source (the workspace) that is not its destination.
template <bool HALVES>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(4, 4))) void ntt_row_inplace_direct_kernel(const PassParams p, const uint32_t rows_total, const uint32_t rows_per_poly, const uint32_t gy) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ldsb[];
    unsigned char *TW1 = ldsb + RowInplaceGeom::XBYTES;
    unsigned char *TW2 = TW1 + RowGeom::TW1_BYTES;
    unsigned char *TW0 = TW2 + RowGeom::TW2_BYTES;   // HALVES only
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned char *X = ldsb + wave * RowInplaceGeom::WBUF_BYTES;   // this wave's buffer

    for (uint32_t e = tid; e < 64 * 16; e += 1024) {
        const uint32_t l = e >> 4, ka = e & 15;
        *reinterpret_cast<uint64_t *>(TW1 + l * RowGeom::TW1_STRIDE + ka * 8) = p.twh[(ka * l) << 2];
        if constexpr (HALVES) *reinterpret_cast<uint64_t *>(TW0 + l * RowInplaceGeom::TW0_STRIDE + ka * 8) = p.twh[(l + 64 * ka) << 1];   // ka plays i here
    }
    if (tid < 64) {
        const uint32_t q = tid >> 4, kb = tid & 15;
        *reinterpret_cast<uint64_t *>(TW2 + q * RowGeom::TW2_STRIDE + kb * 8) = p.twh[(kb * q) << 6];
    }

    // which rows this wave walks: row u of the launch = (z, a, row inside the polynomial), row fastest; HALVES: rows_total counts
    // UNITS (row, half), half fastest — W is even, so a wave keeps its half
    const uint32_t W = gridDim.x * 16;
    const uint32_t u0 = blockIdx.x * 16 + wave;
    const uint32_t n_rows = u0 < rows_total ? (rows_total - u0 + W - 1) / W : 0;
    const uint32_t half = HALVES ? (u0 & 1) : 0;
    auto row_ptr = [&](uint32_t k, bool out) -> uint64_t {
        const uint32_t uu = u0 + k * W, u = HALVES ? uu >> 1 : uu;
        const uint32_t row = u % rows_per_poly, r2 = u / rows_per_poly, a = r2 % gy, z = r2 / gy;
        return out ? a * p.out_sa + z * p.out_sz + (uint64_t)row * p.out_t + (HALVES ? half * 1024u : 0u) : a * p.in_sa + z * p.in_sz + (uint64_t)row * p.in_t;
    };

    const uint32_t q = lane & 3, hi4 = lane >> 2;
    const uint32_t tw1_base = lane * RowGeom::TW1_STRIDE, tw2_base = q * RowGeom::TW2_STRIDE;
    const uint32_t e1w = (hi4 * 68 + q) * 8;        // + kA * 32
    const uint32_t e1r = (hi4 * 4 + q) * 8;         // lane = (kA', q): + h * 544; E2 writes at + kB * 512 with q swizzled
    const uint32_t e2w0 = hi4 * 32;                 // + kB * 512 + ((q ^ (kB >> 2)) * 8)
    auto opaque_lane = [&]() {
        uint32_t l = lane;
        asm volatile("" : "+v"(l));
        return l;
    };

    uint64_t A[16], B[16];
    // HALVES: the other half of the row, x_(j + 1024), arrives in B (free between the last exchange write of a unit and the next first round)
    auto issue_loads = [&](uint32_t k) {
        const uint64_t *base = p.src + row_ptr(k, false);
        const uint32_t off = opaque_lane() * 8;
        static_for<0, 16>([&](auto I_) {
            constexpr int i = decltype(I_)::value;
            A[i] = g_ld(base, off + i * 512);
        });
    };
    // HALVES: the other half of the row of unit k. Issued LATE — behind the last use of B of the unit before it — so that A2 can live in
    // the registers B has just left (A + A2 + B at once do not fit 128 registers); the stores of that unit travel in front of first_round
    auto issue_loads_other_half = [&](uint32_t k) {
        if constexpr (HALVES) {
            const uint64_t *base = p.src + row_ptr(k, false);
            const uint32_t off = opaque_lane() * 8 + 8192;
            static_for<0, 16>([&](auto I_) {
                constexpr int i = decltype(I_)::value;
                B[i] = g_ld(base, off + i * 512);
            });
        }
    };
    // radix 16 over i on A, twiddle, E1 write
    auto first_round = [&]() {
        if constexpr (HALVES) {   // the radix-2 stage of the 2048-point row: this wave's half of its outputs
            if (half == 0) {
                static_for<0, 16>([&](auto I_) { constexpr int i = decltype(I_)::value; A[i] = gl::add(A[i], B[i]); });
            } else {
                // eight elements at a time, fenced: left to itself the scheduler reads all sixteen twiddles ahead (32 more registers)
                static_for<0, 2>([&](auto G_) {
                    constexpr int g0 = 8 * decltype(G_)::value;
                    const uint32_t tw0_base = opaque_lane() * RowInplaceGeom::TW0_STRIDE;
                    static_for<g0, g0 + 8>([&](auto I_) { constexpr int i = decltype(I_)::value; A[i] = gl::sub(A[i], B[i]); });
                    mul_run<g0, g0 + 8>(A, [](auto I_) { return decltype(I_)::value; }, [&](auto I_) { return lds_ld(TW0, tw0_base + decltype(I_)::value * 8); }, [](auto) {});
                    __builtin_amdgcn_sched_barrier(0);
                });
            }
        }
        radix_dif<4, 0>(A);
        mul_run<1, 16>(A, [](auto S_) { return decltype(S_)::value; }, [&](auto S_) { return lds_ld(TW1, tw1_base + brev_c(decltype(S_)::value, 4) * 8); }, [](auto) {});
        static_for<0, 16>([&](auto S_) {
            constexpr int s = decltype(S_)::value;
            constexpr int ka = brev_c(s, 4);
            lds_st(X, e1w + ka * 32, A[s]);
        });
        tile_sync<64>();
    };
    // everything after E1 of the row whose first round is in the buffer, up to its stores
    auto rest_of_row = [&](uint32_t k, bool more) {
        static_for<0, 16>([&](auto H_) {
            constexpr int h = decltype(H_)::value;
            B[h] = lds_ld(X, e1r + h * 544);
        });
        tile_sync<64>();
        radix_dif<4, 0>(B);
        mul_run<1, 16>(B, [](auto S_) { return decltype(S_)::value; }, [&](auto S_) { return lds_ld(TW2, tw2_base + brev_c(decltype(S_)::value, 4) * 8); }, [](auto) {});
        static_for<0, 16>([&](auto S_) {
            constexpr int s = decltype(S_)::value;
            constexpr int kb = brev_c(s, 4);
            lds_st(X, e2w0 + kb * 512 + ((q ^ (uint32_t)(kb >> 2)) * 8), B[s]);
        });
        tile_sync<64>();
        {
            // lane = (kA, kBhi): registers (kBlo, q) from slot 64 (4 kBhi + kBlo) + 4 kA + (q ^ kBhi)
            const uint32_t l = opaque_lane();
            const uint32_t kbhi = l & 3, base = (l >> 2) * 32 + kbhi * 2048;
            static_for<0, 16>([&](auto R_) {
                constexpr int r = decltype(R_)::value;
                B[r] = lds_ld(X, base + (r >> 2) * 512 + (((uint32_t)(r & 3) ^ kbhi) * 8));
            });
        }
        tile_sync<64>();
        radix_dif_blocks<2>(B);
        {
            // E3 write: position = bitrev4(kA) * 64 + bitrev2(kBlo) * 16 + bitrev2(kBhi) * 4 + s2, slot low bits ^ (position >> 8)
            const uint32_t l = opaque_lane();
            const uint32_t ka = l >> 2, kbhi = l & 3;
            const uint32_t pa = brev_rt(ka, 4), sw = pa >> 2;
            const uint32_t base = (pa * 64 + brev_rt(kbhi, 2) * 4) * 8;
            static_for<0, 16>([&](auto R_) {
                constexpr int r = decltype(R_)::value;
                constexpr uint32_t kblo = r >> 2, s2 = r & 3;
                lds_st(X, base + brev_c(kblo, 2) * 128 + ((s2 ^ sw) * 8), gl::canon(B[r]));
            });
        }
        tile_sync<64>();
        {
            const uint32_t l = opaque_lane();
            uint64_t *obase = p.dst + row_ptr(k, true);
            static_for<0, 16>([&](auto R_) {
                constexpr int r = decltype(R_)::value;   // position >> 6
                const uint32_t slot = r * 64 + (l & ~3u) + ((l & 3) ^ (uint32_t)(r >> 2));
                const uint64_t v = lds_ld(X, slot * 8);
                g_st(obase, l * 8 + r * 512, v);
            });
        }
        tile_sync<64>();
        __builtin_amdgcn_sched_barrier(0);   // behind the unit's stores: only the prefetched A is live here
        if (more) issue_loads_other_half(k + 1);
        __builtin_amdgcn_sched_barrier(0);
    };

    lds_barrier();  // tables
    if (n_rows == 0) return;
    issue_loads(0);
    issue_loads_other_half(0);
    first_round();
    if (n_rows > 1) issue_loads(1);
#pragma unroll 1
    for (uint32_t k = 0; k < n_rows; k++) {
        rest_of_row(k, k + 1 < n_rows);
        if (k + 1 < n_rows) {
            first_round();
            if (k + 2 < n_rows) issue_loads(k + 2);
        }
    }
}

template <bool HALVES = false>
hipError_t launch_row_inplace_direct_t(const PassParams &p, dim3 grid, hipStream_t stream) {
    constexpr uint32_t lds_bytes = HALVES ? RowInplaceGeom::LDS_BYTES_HALVES : RowInplaceGeom::LDS_BYTES;
    static DynamicLds attr;
    if (hipError_t e = allow_dynamic_lds(attr, reinterpret_cast<const void *>(&ntt_row_inplace_direct_kernel<HALVES>), lds_bytes); e != hipSuccess) return e;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const uint64_t rows_total = (uint64_t)p.t_limit * grid.y * grid.z * (HALVES ? 2 : 1);
    if (rows_total == 0) return hipSuccess;
    if (rows_total > 0xFFFFFFFFull) return hipErrorInvalidValue;
    const uint64_t need = (rows_total + 15) / 16;
    const uint32_t wgs = (uint32_t)(need < (uint64_t)cus ? need : (uint64_t)cus);
    hipLaunchKernelGGL(ntt_row_inplace_direct_kernel<HALVES>, dim3(wgs), dim3(1024), lds_bytes, stream, p, (uint32_t)rows_total, p.t_limit, (uint32_t)grid.y);
    return hipGetLastError();
}

}  // namespace

// Can the coset column pass run as a direct pass? The workgroup must keep one column tile (as many column tiles as CUs at most) and
// the per-coset tables must fit beside the exchange image.
bool col_direct_coset_ok(int logg, dim3 grid) {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (grid.x > (uint32_t)cus || grid.z == 0) return false;
    const uint32_t limit = device_lds_limit();
    switch (logg) {
        case 0: return ColGeom<0>::LDS_BYTES + ColGeom<0>::coset_bytes(grid.z) <= limit;
        case 1: return ColGeom<1>::LDS_BYTES + ColGeom<1>::coset_bytes(grid.z) <= limit;
        case 2: return ColGeom<2>::LDS_BYTES + ColGeom<2>::coset_bytes(grid.z) <= limit;
        default: return false;
    }
}

hipError_t launch_col_direct(int logg, const PassParams &p, dim3 grid, hipStream_t stream) {
    const bool nat = p.flags & F_NATURAL;
    if (p.flags & F_COSET) {
        if (nat || !col_direct_coset_ok(logg, grid)) return hipErrorInvalidValue;
        switch (logg) {
            case 0: return launch_col_direct_t<0, false, true>(p, grid, stream);
            case 1: return launch_col_direct_t<1, false, true>(p, grid, stream);
            case 2: return launch_col_direct_t<2, false, true>(p, grid, stream);
            default: return hipErrorInvalidValue;
        }
    }
    switch (logg) {
        case 0: return nat ? launch_col_direct_t<0, true>(p, grid, stream) : launch_col_direct_t<0, false>(p, grid, stream);
        case 1: return nat ? launch_col_direct_t<1, true>(p, grid, stream) : launch_col_direct_t<1, false>(p, grid, stream);
        case 2: return nat ? launch_col_direct_t<2, true>(p, grid, stream) : launch_col_direct_t<2, false>(p, grid, stream);
        case 3: return nat ? launch_col_direct_t<3, true>(p, grid, stream) : launch_col_direct_t<3, false>(p, grid, stream);   // 2048-point columns, 64-byte segments
        default: return hipErrorInvalidValue;
    }
}


// eight lane groups, natural order, the input read index-reversed (see the kernel): first pass of the inverse 2^22 two-pass plan
hipError_t launch_col_direct_reversed_input(const PassParams &p, dim3 grid, hipStream_t stream) {
    if (!(p.flags & F_NATURAL) || (p.flags & F_COSET) || p.in_t != 1 || p.in_m != (1u << 11) || p.in_sb != 8) return hipErrorInvalidValue;
    return launch_col_direct_t<3, true, false, false, true>(p, grid, stream);
}

hipError_t launch_col_direct_final(const PassParams &p, dim3 grid, hipStream_t stream) {
    if (!(p.flags & F_NATURAL) || (p.flags & F_COSET)) return hipErrorInvalidValue;
    return launch_col_direct_t<3, true, false, true>(p, grid, stream);
}

hipError_t launch_row_inplace_direct(const PassParams &p, dim3 grid, hipStream_t stream) { return launch_row_inplace_direct_t<false>(p, grid, stream); }
// rows of 2048 points in place, two waves per row (the kernel's HALVES form)
hipError_t launch_row_inplace_direct_2048(const PassParams &p, dim3 grid, hipStream_t stream) { return launch_row_inplace_direct_t<true>(p, grid, stream); }

hipError_t launch_row_natural_direct(const PassParams &p, dim3 grid, hipStream_t stream) {
    return (p.flags & F_INVERSE) ? launch_row_natural_direct_t<true>(p, grid, stream) : launch_row_natural_direct_t<false>(p, grid, stream);
}

}  // namespace nttk
}  // namespace plonky2_hip
