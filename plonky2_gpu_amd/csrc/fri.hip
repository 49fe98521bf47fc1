// fri.hip — device primitives of the FRI opening pipeline:
//   reduce_polys_base     ReducingFactor::reduce_polys_base (plonky2/src/util/reducing.rs:83-95)
//   divide_by_linear      PolynomialCoeffs::divide_by_linear (field/src/polynomial/division.rs:75-88),
//                         accumulated into final_poly as in prove_openings (fri/oracle.rs:1069-1087)
//   fold                  the per-layer coefficient folding of fri_committed_trees (fri/prover.rs:103-111)
//   interleave            flatten() of extension values into Merkle leaves (fri/prover.rs:90-95)
//   proof of work         fri_proof_of_work (fri/prover.rs:122-171), smallest witness
// Extension elements a + bX (X^2 = 7) are kept PLANAR on the device: plane 0 = all a, plane 1 = all b,
// because the NTT acts on the two components independently with base-field twiddles, so every
// transform of an extension polynomial is two columns of the batched base NTT.
//
// divide_by_linear is a Horner recurrence q_{i-1} = c_i + z q_i. Written as
//   q_i = z^-(i+1) * sum_{k>i} c_k z^k
// it becomes: scale by z^k, SUFFIX SUM (additions only — a plain parallel scan), scale by z^-(i+1).
#include "fri.h"

#include "poseidon.h"

namespace plonky2_hip {
namespace {

struct Ext2 {
    uint64_t a, b;
};

__device__ __forceinline__ Ext2 ext_mul(Ext2 x, Ext2 y) {
    uint64_t t = gl::mul(x.b, y.b);
    uint64_t t7 = gl::sub(gl::mul_pow2<3>(t), t);  // W = 7 (field/src/goldilocks_extensions.rs:19)
    return Ext2{gl::add(gl::mul(x.a, y.a), t7), gl::mac(gl::mul(x.a, y.b), x.b, y.a)};
}
__device__ __forceinline__ Ext2 ext_add(Ext2 x, Ext2 y) { return Ext2{gl::add(x.a, y.a), gl::add(x.b, y.b)}; }
__device__ __forceinline__ Ext2 ext_pow(Ext2 base, uint64_t e) {
    Ext2 acc{1, 0};
    while (e) {
        if (e & 1) acc = ext_mul(acc, base);
        base = ext_mul(base, base);
        e >>= 1;
    }
    return acc;
}
__device__ __forceinline__ Ext2 ext_inv(Ext2 x) {
    // 1/(a + bX) = (a - bX) / (a^2 - 7 b^2)
    uint64_t b2 = gl::sqr(x.b);
    uint64_t d = gl::sub(gl::sqr(x.a), gl::sub(gl::mul_pow2<3>(b2), b2));
    uint64_t di = gl::pow(d, gl::P - 2);
    return Ext2{gl::mul(x.a, di), gl::mul(gl::neg(x.b), di)};
}

unsigned grid_for(uint64_t n, unsigned block) { return (unsigned)((n + block - 1) / block); }

// pw[j] = alpha^j, j < m   (one thread: m is a few hundred)
__global__ void ext_powers_kernel(Ext2 alpha, uint32_t m, uint64_t *pw) {
    if (blockIdx.x || threadIdx.x) return;
    Ext2 p{1, 0};
    for (uint32_t j = 0; j < m; j++) {
        pw[2 * j] = gl::canon(p.a);
        pw[2 * j + 1] = gl::canon(p.b);
        p = ext_mul(p, alpha);
    }
}

// the builtin returns int: go through uint32_t or the low word sign-extends over the high one
__device__ __forceinline__ uint64_t uniform64(uint64_t v) {
    uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
    uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    return (uint64_t)lo | ((uint64_t)hi << 32);
}

// out = sum_j alpha^j * poly_j  (base polynomials, extension result, planar)
__global__ __launch_bounds__(256) void reduce_polys_base_kernel(const uint64_t *const *__restrict__ polys, uint32_t m,
                                                                const uint64_t *__restrict__ pw, uint64_t n, uint64_t *out) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    gl::DotAcc a, b;
    for (uint32_t j = 0; j < m; j++) {
        uint64_t c = polys[j][i];
        uint64_t pa = uniform64(pw[2 * j]), pb = uniform64(pw[2 * j + 1]);
        // the hazard recognizer does not look inside inline asm: v_readfirstlane (VALU writes SGPR) must be
        // two wait states away from dot_term's first VALU read of that SGPR
        asm volatile("s_nop 2" : "+s"(pa), "+s"(pb));
        gl::dot_term(a, c, pa);
        gl::dot_term(b, c, pb);
    }
    out[i] = gl::canon(gl::dot_finish(a));
    out[n + i] = gl::canon(gl::dot_finish(b));
}

// two-level power tables of z: lo[e] = z^e (e < 1024), hi[e] = z^(1024 e) (e < hi_len); 2 u64 per entry
__global__ void ext_pow_tables_kernel(Ext2 z, uint32_t hi_len, uint64_t *lo, uint64_t *hi) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 1024 + hi_len) return;
    Ext2 r = i < 1024 ? ext_pow(z, i) : ext_pow(z, (uint64_t)(i - 1024) << 10);
    uint64_t *o = i < 1024 ? lo + 2 * i : hi + 2 * (i - 1024);
    o[0] = gl::canon(r.a);
    o[1] = gl::canon(r.b);
}

__device__ __forceinline__ Ext2 table_pow(const uint64_t *lo, const uint64_t *hi, uint64_t e) {
    const uint64_t *l = lo + 2 * (e & 1023), *h = hi + 2 * (e >> 10);
    return ext_mul(Ext2{h[0], h[1]}, Ext2{l[0], l[1]});
}

// v[k] *= z^k (planar)
__global__ __launch_bounds__(256) void ext_scale_powers_kernel(uint64_t *v, uint64_t n, const uint64_t *lo, const uint64_t *hi) {
    uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    Ext2 r = ext_mul(Ext2{v[k], v[n + k]}, table_pow(lo, hi, k));
    v[k] = r.a;
    v[n + k] = r.b;
}

// ---- reverse (suffix) inclusive sum of a planar extension vector, three kernels ----------------
constexpr int SC_T = 256, SC_E = 4, SC_B = SC_T * SC_E;

__device__ __forceinline__ uint64_t block_inclusive_add(uint64_t v, uint64_t *lds) {
    const uint32_t t = threadIdx.x;
    lds[t] = v;
    __syncthreads();
    for (int off = 1; off < SC_T; off <<= 1) {
        uint64_t x = (t >= (uint32_t)off) ? gl::add(lds[t - off], lds[t]) : lds[t];
        __syncthreads();
        lds[t] = x;
        __syncthreads();
    }
    uint64_t r = lds[t];
    __syncthreads();
    return r;
}

// position r counts from the END: element index = n-1-r. In place: v <- suffix sums within the block.
__global__ __launch_bounds__(SC_T) void suffix_blocks_kernel(uint64_t *v, uint64_t n, uint64_t *totals, uint64_t n_blocks) {
    __shared__ uint64_t lds[SC_T];
    uint64_t *plane = v + (uint64_t)blockIdx.y * n;
    uint64_t r0 = (uint64_t)blockIdx.x * SC_B + (uint64_t)threadIdx.x * SC_E;
    uint64_t e[SC_E], s = 0;
#pragma unroll
    for (int k = 0; k < SC_E; k++) {
        e[k] = r0 + k < n ? plane[n - 1 - (r0 + k)] : 0;
        s = gl::add(s, e[k]);
    }
    uint64_t incl = block_inclusive_add(s, lds);
    uint64_t run = gl::sub(incl, s);  // exclusive prefix of this thread
#pragma unroll
    for (int k = 0; k < SC_E; k++) {
        run = gl::add(run, e[k]);
        if (r0 + k < n) plane[n - 1 - (r0 + k)] = run;
    }
    if (threadIdx.x == SC_T - 1) totals[(uint64_t)blockIdx.y * n_blocks + blockIdx.x] = incl;
}

__global__ __launch_bounds__(SC_T) void suffix_totals_kernel(uint64_t *totals, uint64_t m) {
    __shared__ uint64_t lds[SC_T];
    uint64_t *t = totals + (uint64_t)blockIdx.x * m;
    uint64_t per = (m + SC_T - 1) / SC_T;
    uint64_t lo = (uint64_t)threadIdx.x * per, hi = lo + per < m ? lo + per : m;
    uint64_t s = 0;
    for (uint64_t i = lo; i < hi; i++) s = gl::add(s, t[i]);
    uint64_t run = gl::sub(block_inclusive_add(s, lds), s);
    for (uint64_t i = lo; i < hi; i++) {  // exclusive prefix over blocks
        uint64_t x = t[i];
        t[i] = run;
        run = gl::add(run, x);
    }
}

// final[i+1] = final[i+1]*scale + z^-(i+1) * S_{i+1}, S = suffix sums (block-local + block prefix); final[0] = 0
__global__ __launch_bounds__(256) void divide_finish_kernel(const uint64_t *suf, const uint64_t *totals, uint64_t n_blocks, uint64_t n,
                                                            const uint64_t *ilo, const uint64_t *ihi, Ext2 scale, int accumulate,
                                                            uint64_t *fin) {
    uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;  // j = i+1 in 1..n-1, plus j = 0
    if (j >= n) return;
    if (j == 0) {
        fin[0] = 0;
        fin[n] = 0;
        return;
    }
    uint64_t blk = (n - 1 - j) / SC_B;
    Ext2 S{gl::add(suf[j], totals[blk]), gl::add(suf[n + j], totals[n_blocks + blk])};
    Ext2 q = ext_mul(S, table_pow(ilo, ihi, j));
    if (accumulate) q = ext_add(q, ext_mul(Ext2{fin[j], fin[n + j]}, scale));
    fin[j] = gl::canon(q.a);
    fin[n + j] = gl::canon(q.b);
}

// out[k] = sum_{i < arity} c[k*arity + i] * beta^i   (reduce_with_powers, plonk_common.rs:116-128)
// d_beta != null: beta is read where the device-resident transcript left it (two canonical words)
__global__ __launch_bounds__(256) void fold_kernel(const uint64_t *c, uint64_t len, uint32_t arity_bits, Ext2 beta, uint64_t *out,
                                                   const uint64_t *__restrict__ d_beta) {
    uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t out_len = len >> arity_bits;
    if (k >= out_len) return;
    if (d_beta) beta = Ext2{d_beta[0], d_beta[1]};
    Ext2 s{0, 0};
    for (int64_t i = (1ll << arity_bits) - 1; i >= 0; i--) {
        uint64_t idx = (k << arity_bits) + i;
        s = ext_add(ext_mul(s, beta), Ext2{c[idx], c[len + idx]});
    }
    out[k] = gl::canon(s.a);
    out[out_len + k] = gl::canon(s.b);
}

// rows[2 idx + c] = plane_c[idx]
__global__ __launch_bounds__(256) void interleave_kernel(const uint64_t *planes, uint64_t len, uint64_t *rows) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= len) return;
    rows[2 * i] = planes[i];
    rows[2 * i + 1] = planes[len + i];
}

struct PowState {
    uint64_t s[12];
};

// candidates base .. base + count: smallest one whose response has enough leading zeros (atomicMin)
// T != null: the transcript lives on the device (merkle.hip challenger_step_kernel): state T[0..12) with the input buffer T[12..) of
// length T[28] written over its first words, the candidate behind it (fri/prover.rs:137-147)
__global__ __launch_bounds__(256) void pow_kernel(PowState st, uint32_t pos, uint32_t min_leading_zeros, uint64_t base, uint64_t count,
                                                  unsigned long long *best, const uint64_t *__restrict__ T) {
    const poseidon::MdsOperands ops = poseidon::mds_operands();
    uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = g < count;  // whole waves to the end (poseidon.h): lanes past the end try the last candidate once more
    if (!live) g = count - 1;
    uint64_t cand = base + g;
    uint64_t s[12];
    if (T) {
        pos = (uint32_t)T[28];
#pragma unroll
        for (int k = 0; k < 12; k++) s[k] = gl::canon(((uint32_t)k < pos && k < 8) ? T[12 + k] : T[k]);
    } else {
#pragma unroll
        for (int k = 0; k < 12; k++) s[k] = st.s[k];
    }
#pragma unroll
    for (int k = 0; k < 12; k++)
        if ((uint32_t)k == pos) s[k] = cand;
    poseidon::permute(s, ops);
    uint64_t resp = gl::canon(s[7]);  // duplex_state[SPONGE_RATE - 1]
    uint32_t lz = resp ? (uint32_t)__clzll((long long)resp) : 64u;
    if (lz >= min_leading_zeros) atomicMin(best, (unsigned long long)cand);
}

}  // namespace

hipError_t fri_reduce_polys_base(const NttTables &tb, const uint64_t *const *d_poly_ptrs, uint32_t m, uint64_t n, const uint64_t alpha[2],
                                 uint64_t *d_out, hipStream_t stream) {
    if (m == 0 || !tb.scratch || tb.scratch_elems < 2ull * m) return hipErrorInvalidValue;
    uint64_t *pw = tb.scratch;
    hipLaunchKernelGGL(ext_powers_kernel, dim3(1), dim3(64), 0, stream, Ext2{alpha[0] % glh::P, alpha[1] % glh::P}, m, pw);
    hipLaunchKernelGGL(reduce_polys_base_kernel, dim3(grid_for(n, 256)), dim3(256), 0, stream, d_poly_ptrs, m, pw, n, d_out);
    return hipGetLastError();
}

hipError_t fri_divide_by_linear_accumulate(const NttTables &tb, uint64_t *d_comp, uint64_t n, const uint64_t z[2], const uint64_t scale[2],
                                           int accumulate, uint64_t *d_final, hipStream_t stream) {
    if (n < 2) return hipErrorInvalidValue;
    const uint32_t hi_len = (uint32_t)((n + 1023) >> 10) + 1;
    const uint64_t n_blocks = (n + SC_B - 1) / SC_B;
    // workspace: z tables, z^-1 tables, block totals
    const uint64_t tbl = 2ull * (1024 + hi_len);
    if (!tb.scratch || tb.scratch_elems < 2 * tbl + 2 * n_blocks) return hipErrorInvalidValue;
    uint64_t *lo = tb.scratch, *hi = lo + 2 * 1024, *ilo = tb.scratch + tbl, *ihi = ilo + 2 * 1024, *totals = tb.scratch + 2 * tbl;
    const Ext2 zz{z[0] % glh::P, z[1] % glh::P};
    // z^-1 on the host: 1/(a + bX) = (a - bX)/(a^2 - 7 b^2)
    uint64_t d = glh::add(glh::mul(zz.a, zz.a), glh::P - glh::mul(7, glh::mul(zz.b, zz.b)));
    if (d == 0) return hipErrorInvalidValue;
    uint64_t di = glh::inv(d);
    const Ext2 zi{glh::mul(zz.a, di), glh::mul(zz.b ? glh::P - zz.b : 0, di)};
    hipLaunchKernelGGL(ext_pow_tables_kernel, dim3(grid_for(1024 + hi_len, 256)), dim3(256), 0, stream, zz, hi_len, lo, hi);
    hipLaunchKernelGGL(ext_pow_tables_kernel, dim3(grid_for(1024 + hi_len, 256)), dim3(256), 0, stream, zi, hi_len, ilo, ihi);
    hipLaunchKernelGGL(ext_scale_powers_kernel, dim3(grid_for(n, 256)), dim3(256), 0, stream, d_comp, n, lo, hi);
    hipLaunchKernelGGL(suffix_blocks_kernel, dim3((unsigned)n_blocks, 2), dim3(SC_T), 0, stream, d_comp, n, totals, n_blocks);
    hipLaunchKernelGGL(suffix_totals_kernel, dim3(2), dim3(SC_T), 0, stream, totals, n_blocks);
    hipLaunchKernelGGL(divide_finish_kernel, dim3(grid_for(n, 256)), dim3(256), 0, stream, d_comp, totals, n_blocks, n, ilo, ihi,
                       Ext2{scale[0] % glh::P, scale[1] % glh::P}, accumulate, d_final);
    return hipGetLastError();
}

hipError_t fri_fold(const uint64_t *d_coeffs, uint64_t len, uint32_t arity_bits, const uint64_t beta[2], uint64_t *d_out,
                    hipStream_t stream, const uint64_t *d_beta) {
    if (arity_bits == 0 || arity_bits > 8 || (len >> arity_bits) == 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(fold_kernel, dim3(grid_for(len >> arity_bits, 256)), dim3(256), 0, stream, d_coeffs, len, arity_bits,
                       d_beta ? Ext2{0, 0} : Ext2{beta[0] % glh::P, beta[1] % glh::P}, d_out, d_beta);
    return hipGetLastError();
}

hipError_t fri_interleave(const uint64_t *d_planes, uint64_t len, uint64_t *d_rows, hipStream_t stream) {
    if (len == 0) return hipSuccess;
    hipLaunchKernelGGL(interleave_kernel, dim3(grid_for(len, 256)), dim3(256), 0, stream, d_planes, len, d_rows);
    return hipGetLastError();
}

hipError_t fri_proof_of_work(const NttTables &tb, const uint64_t state[12], uint32_t pos, uint32_t min_leading_zeros, uint64_t *witness,
                             hipStream_t stream, const uint64_t *d_challenger, uint64_t *d_witness) {
    if (pos >= 12 || !tb.scratch) return hipErrorInvalidValue;
    PowState st = {};
    if (!d_challenger)
        for (int k = 0; k < 12; k++) st.s[k] = state[k] % glh::P;
    unsigned long long *best = d_witness ? reinterpret_cast<unsigned long long *>(d_witness) : reinterpret_cast<unsigned long long *>(tb.scratch);
    // Batches are scanned in order, so the first batch that holds a witness holds the smallest one. The
    // expected search length is 2^min_leading_zeros' (16 proof-of-work bits -> 2^16 candidates): the batch
    // starts at 2^17 (one launch ~ one permutation latency) and doubles while nothing is found.
    uint64_t batch = 1ull << 17;
    for (uint64_t base = 0; base < glh::P; base += batch, batch = batch < (1ull << 24) ? batch * 2 : batch) {
        hipError_t e = hipMemsetAsync(best, 0xFF, 8, stream);
        if (e != hipSuccess) return e;
        const uint64_t this_batch = batch;
        uint64_t count = glh::P - base < this_batch ? glh::P - base : this_batch;
        hipLaunchKernelGGL(pow_kernel, dim3(grid_for(count, 256)), dim3(256), 0, stream, st, pos, min_leading_zeros, base, count, best, d_challenger);
        unsigned long long h = ~0ull;
        e = hipMemcpyAsync(&h, best, 8, hipMemcpyDeviceToHost, stream);
        if (e != hipSuccess) return e;
        e = hipStreamSynchronize(stream);
        if (e != hipSuccess) return e;
        if (h != ~0ull) {
            *witness = h;
            return hipSuccess;
        }
    }
    return hipErrorUnknown;
}

}  // namespace plonky2_hip
