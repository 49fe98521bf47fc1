// ntt.h — internal interface of the NTT / coset-LDE planner (see ntt.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace plonky2_hip {

constexpr uint32_t NTT_MAX_LOG = 24;

// Device-resident twiddle tables shared by every transform size (64 KiB, L2-resident):
//   twl[e] = w_{2^24}^e, twh[e] = w_{2^12}^e, e < 4096, so w_{2^24}^x = twh[x>>12] * twl[x&4095].
struct NttTables {
    uint64_t *twl = nullptr;
    uint64_t *twh = nullptr;
    // Workspace for natural-order multi-pass transforms (the last pass writes transposed, so the
    // intermediate cannot live in the caller's buffer), 512 MiB = 64 columns of 2^20: one launch pair
    // per batch. One per CONTEXT (capi.hip CtxState holds a copy of the device's tables with its own
    // scratch), so that two contexts on one device never meet in it.
    uint64_t *scratch = nullptr;
    uint64_t scratch_elems = 0;
};
constexpr uint64_t NTT_SCRATCH_ELEMS = 1ull << 26;

// Per-(log_n, rate_bits, shift) coset tables: s_r = shift * w_{n<<rate_bits}^r,
//   lo[r*1024 + e] = s_r^e (e < 1024), hi[r*hi_len + e] = s_r^(1024 e) (e < hi_len = max(n/1024, 1)).
struct CosetTables {
    uint64_t *lo = nullptr;
    uint64_t *hi = nullptr;
    uint32_t hi_len = 0;
    uint32_t log_n = 0, rate_bits = 0;
    uint64_t shift = 0;
};

enum class NttOrder { Natural, BitReversed };

hipError_t ntt_tables_create(NttTables *tb);
void ntt_tables_destroy(NttTables *tb);
hipError_t coset_tables_create(CosetTables *ct, uint32_t log_n, uint32_t rate_bits, uint64_t shift, hipStream_t stream);
void coset_tables_destroy(CosetTables *ct);

// Forward (or inverse) NTT of n_polys polynomials of length 2^log_n; polynomial i lives at
// src + i*src_stride and is written to dst + i*dst_stride (src == dst allowed when the strides
// match). Natural order = fft_with_options / ifft_with_options (field/src/fft.rs:58-103);
// BitReversed = position m holds frequency bitrev(m) (forward only).
hipError_t ntt_batch(const NttTables &tb, const uint64_t *src, uint64_t *dst, uint64_t n_polys, uint32_t log_n,
                     uint64_t src_stride, uint64_t dst_stride, NttOrder order, bool inverse, hipStream_t stream);

// Coset low-degree extension (PolynomialCoeffs::lde + coset_fft_with_options,
// field/src/polynomial/mod.rs:205-207, 286-299) of n_polys coefficient vectors of length 2^log_n
// to 2^(log_n+rate_bits) evaluations on shift*H, written in BIT-REVERSED order (leaf order,
// fri/oracle.rs:942-952): dst[i*dst_stride + m] = evaluation at natural index bitrev(m).
hipError_t coset_lde_batch(const NttTables &tb, const CosetTables &ct, const uint64_t *coeffs, uint64_t *dst,
                           uint64_t n_polys, uint64_t src_stride, uint64_t dst_stride, hipStream_t stream);

// values[poly*stride + i] *= s^i with s = ct.shift (ct built with rate_bits = 0).
hipError_t scale_by_powers(const CosetTables &ct, uint64_t *values, uint64_t n_polys, uint64_t stride, hipStream_t stream);

}  // namespace plonky2_hip
