// TEST INFRASTRUCTURE ONLY -- never linked into, loaded by, or timed as the product.
//
// oracle/_ref: the reference's OWN device kernels (sideprotocol/plonky2-gpu, cuda/plonky2_gpu_impl.cuh with
// cuda/def.cuh, cuda/constants.cuh and cuda/*Gate.cuh) compiled for gfx950 from the sources where they lie under
// /root/reference -- no edit to, no copy of and no stand-in for any reference file; the recipe is oracle/Makefile's
// `_ref` target (hipcc -include hip/hip_runtime.h -Wno-c++11-narrowing -I/root/reference/cuda).  What is mine in this
// file is only the host side: each entry point below launches the reference kernels with the launch geometry of the
// reference's host code (cuda/plonky2_gpu.cu, cited per launch) on buffers the caller owns, and times them with HIP events.
// The reference's host file itself is CUDA-runtime code (cudaStream_t, <<<>>> on cudaStreams, RustError) and is not compiled.
//
// The kernels are the reference's GPU twin of its Rust CPU prover -- the code its authors shipped proofs with -- so
// tests/test_gpu_reference_kernels.py uses them as an independent source of truth for the product's NTT/LDE, Merkle and
// quotient rows, and bench.py's reference leg reports their durations on the same MI355X (a stated baseline, never the target).
#include "plonky2_gpu_impl.cuh"

#include <hip/hip_runtime.h>

namespace {

struct Timer {
    hipEvent_t a, b;
    float* out;
    explicit Timer(float* ms) : out(ms) {
        hipEventCreate(&a);
        hipEventCreate(&b);
        hipEventRecord(a, 0);
    }
    int stop() {
        hipEventRecord(b, 0);
        hipError_t e = hipEventSynchronize(b);
        if (e == hipSuccess) e = hipGetLastError();
        float ms = 0.f;
        if (e == hipSuccess) hipEventElapsedTime(&ms, a, b);
        if (out) *out = ms;
        hipEventDestroy(a);
        hipEventDestroy(b);
        return (int)e;
    }
};

inline GoldilocksField* gf(uint64_t* p) { return reinterpret_cast<GoldilocksField*>(p); }
inline const GoldilocksField* gf(const uint64_t* p) { return reinterpret_cast<const GoldilocksField*>(p); }
inline unsigned blocks32(long thcnt) { return (unsigned)((thcnt + 31) / 32); }

// The quotient kernel calls the gates through member-function pointers (cuda/plonky2_gpu_impl.cuh:619-630, 741-743), so its
// stack is "dynamic" for the AMD code generator: the code object records only the kernel's own frame (5952 bytes per lane) and
// the HIP runtime gives a dynamic-stack kernel max(that, hipLimitStackSize) -- 1 KiB by default, i.e. nothing for the callees, whose
// frames (RandomAccessGate: 1984 bytes) then land in the next wave's scratch: harmless with a handful of waves, a memory fault
// at 2^11 points. 16 KiB covers the kernel's frame plus the deepest gate with room to spare.
inline void quotient_stack() {
    static bool done = false;
    if (!done) {
        hipDeviceSetLimit(hipLimitStackSize, 16384);
        done = true;
    }
}

}  // namespace

extern "C" {

const char* ref_error_string(int code) { return hipGetErrorString((hipError_t)code); }

// ifft (cuda/plonky2_gpu.cu:70-86): one 256-thread block per polynomial (:81)
int ref_ifft(uint64_t* d_values, int poly_num, int n, int log_n, const uint64_t* d_root_table, uint64_t n_inv, float* ms) {
    Timer t(ms);
    ifft_kernel<<<poly_num, 32 * 8, 0, 0>>>(gf(d_values), poly_num, n, log_n, gf(d_root_table), GoldilocksField{n_inv});
    return t.stop();
}

// the LDE of merkle_tree_from_coeffs (cuda/plonky2_gpu.cu:481-521): lde_kernel :484, init_lde_kernel :491,
// mul_shift_kernel :498, fft_kernel with r = rate_bits :519.  d_ext is the column-major region ("region B").
// ms[0..3] = the four kernels.
int ref_coset_lde(const uint64_t* d_coeffs, uint64_t* d_ext, int poly_num, int n, int log_n, const uint64_t* d_root_table2,
                  const uint64_t* d_shift_powers, int rate_bits, float* ms) {
    long thcnt = (long)n * poly_num;
    int e;
    {
        Timer t(ms ? ms + 0 : nullptr);
        lde_kernel<<<blocks32(thcnt), 32, 0, 0>>>(gf(d_coeffs), gf(d_ext), poly_num, n, rate_bits);
        if ((e = t.stop())) return e;
    }
    {
        Timer t(ms ? ms + 1 : nullptr);
        init_lde_kernel<<<blocks32(thcnt), 32, 0, 0>>>(gf(d_ext), poly_num, n, rate_bits);
        if ((e = t.stop())) return e;
    }
    {
        Timer t(ms ? ms + 2 : nullptr);
        mul_shift_kernel<<<blocks32(thcnt), 32, 0, 0>>>(gf(d_ext), poly_num, n, rate_bits, gf(d_shift_powers));
        if ((e = t.stop())) return e;
    }
    {
        Timer t(ms ? ms + 3 : nullptr);
        fft_kernel<<<poly_num, 32 * 8, 0, 0>>>(gf(d_ext), poly_num, n << rate_bits, log_n + rate_bits, gf(d_root_table2), rate_bits);
        if ((e = t.stop())) return e;
    }
    return 0;
}

// plain forward transform, natural in -> natural out: fft_kernel with r = 0 (the same kernel, cuda/plonky2_gpu_impl.cuh:254)
int ref_fft(uint64_t* d_values, int poly_num, int n, int log_n, const uint64_t* d_root_table, int r, float* ms) {
    Timer t(ms);
    fft_kernel<<<poly_num, 32 * 8, 0, 0>>>(gf(d_values), poly_num, n, log_n, gf(d_root_table), r);
    return t.stop();
}

// reverse_index_bits_kernel as launched at cuda/plonky2_gpu.cu:543-545 (one thread per element)
int ref_reverse_index_bits(uint64_t* d_ext, int poly_num, int n_ext, int log_n_ext, float* ms) {
    Timer t(ms);
    reverse_index_bits_kernel<<<blocks32((long)n_ext * poly_num), 32, 0, 0>>>(gf(d_ext), poly_num, n_ext, log_n_ext);
    return t.stop();
}

// hash_leaves_kernel :557 + reduce_digests_kernel :563-565 over column-major d_ext ([leaf_len][n_ext], already in leaf order);
// digests||cap are written behind the columns, at d_ext + n_ext*leaf_len (:552).  ms[0..1].
int ref_merkle_tree(uint64_t* d_ext, int leaf_len, int n_ext, int cap_height, float* ms) {
    int len_cap = 1 << cap_height;
    int num_digests = 2 * (n_ext - len_cap);
    auto* d_digest_buf = (PoseidonHasher::HashOut*)(gf(d_ext) + (size_t)n_ext * leaf_len);
    int e;
    {
        Timer t(ms ? ms + 0 : nullptr);
        hash_leaves_kernel<<<blocks32(n_ext), 32, 0, 0>>>(gf(d_ext), leaf_len, n_ext, d_digest_buf, len_cap, num_digests);
        if ((e = t.stop())) return e;
    }
    {
        Timer t(ms ? ms + 1 : nullptr);
        int nthreads = 32 * 8;
        reduce_digests_kernel<<<len_cap, nthreads, 0, 0>>>(n_ext, d_digest_buf, len_cap, num_digests);
        if ((e = t.stop())) return e;
    }
    return 0;
}

// transpose_kernel as launched at cuda/plonky2_gpu.cu:589-591: column-major [poly_num][n_ext] -> leaf-major [n_ext][poly_num]
int ref_transpose(const uint64_t* d_src, uint64_t* d_dst, int poly_num, int n_ext, float* ms) {
    Timer t(ms);
    transpose_kernel<<<blocks32(n_ext), 32, 0, 0>>>(const_cast<GoldilocksField*>(gf(d_src)), gf(d_dst), poly_num, n_ext);
    return t.stop();
}

// compute_quotient_polys (cuda/plonky2_gpu.cu:609-783): compute_quotient_values_kernel :690 with 300000 threads in
// blocks of 32 (:684-685) and the circuit constants of :665-673, transpose_kernel :741, ifft_kernel :747 (two blocks of 256),
// mul_kernel :762.  The public-inputs hash, hard-wired at :686-689 for the authors' one proof, is an argument here;
// n_inv_ext (hard-wired at :746 for 2^21 points) likewise, so that smaller sizes can be compared too.  ms[0..3].
int ref_compute_quotient_polys(uint64_t* d_wires_leaves, int log_len, int rate_bits, uint64_t* d_zs_pp_leaves, uint64_t* d_const_sigma_leaves,
                               uint64_t* d_outs, uint64_t* d_quotient_polys, uint64_t* d_points, uint64_t* d_z_h_evals,
                               uint64_t* d_z_h_inverses, uint64_t* d_k_is, uint64_t* d_alphas, uint64_t* d_betas, uint64_t* d_gammas,
                               const uint64_t* d_root_table2, const uint64_t* d_shift_inv_powers, const uint64_t* public_inputs_hash,
                               uint64_t n_inv_ext, float* ms) {
    int n_ext = 1 << (log_len + rate_bits);
    int num_challenges = 2, num_gate_constraints = 231, num_constants = 8, num_routed_wires = 80, quotient_degree_factor = 8,
        num_partial_products = 9, cs_leaf_len = 88, zs_leaf_len = 20, wires_leaf_len = 234;
    PoseidonHasher::HashOut pih = {GoldilocksField{public_inputs_hash[0]}, GoldilocksField{public_inputs_hash[1]},
                                   GoldilocksField{public_inputs_hash[2]}, GoldilocksField{public_inputs_hash[3]}};
    int e;
    quotient_stack();
    {
        Timer t(ms ? ms + 0 : nullptr);
        int thcnt = 300000;
        compute_quotient_values_kernel<<<blocks32(thcnt), 32, 0, 0>>>(
            log_len, rate_bits, gf(d_points), gf(d_outs), pih, gf(d_const_sigma_leaves), cs_leaf_len, gf(d_zs_pp_leaves), zs_leaf_len,
            gf(d_wires_leaves), wires_leaf_len, num_constants, num_routed_wires, num_challenges, num_gate_constraints,
            quotient_degree_factor, num_partial_products, gf(d_z_h_evals), gf(d_z_h_inverses), gf(d_k_is), gf(d_alphas), gf(d_betas),
            gf(d_gammas));
        if ((e = t.stop())) return e;
    }
    {
        Timer t(ms ? ms + 1 : nullptr);
        transpose_kernel<<<blocks32(n_ext), 32, 0, 0>>>(gf(d_outs), gf(d_quotient_polys), n_ext, num_challenges);
        if ((e = t.stop())) return e;
    }
    {
        Timer t(ms ? ms + 2 : nullptr);
        ifft_kernel<<<num_challenges, 32 * 8, 0, 0>>>(gf(d_quotient_polys), num_challenges, n_ext, log_len + rate_bits, gf(d_root_table2),
                                                      GoldilocksField{n_inv_ext});
        if ((e = t.stop())) return e;
    }
    {
        Timer t(ms ? ms + 3 : nullptr);
        mul_kernel<<<blocks32((long)n_ext * num_challenges), 32, 0, 0>>>(gf(d_quotient_polys), num_challenges, n_ext, gf(d_shift_inv_powers));
        if ((e = t.stop())) return e;
    }
    return 0;
}

// only the per-point values (outs[2*index + challenge], before the transposition and the coset iFFT)
int ref_compute_quotient_values(uint64_t* d_wires_leaves, int log_len, int rate_bits, uint64_t* d_zs_pp_leaves, uint64_t* d_const_sigma_leaves,
                                uint64_t* d_outs, uint64_t* d_points, uint64_t* d_z_h_evals, uint64_t* d_z_h_inverses, uint64_t* d_k_is,
                                uint64_t* d_alphas, uint64_t* d_betas, uint64_t* d_gammas, const uint64_t* public_inputs_hash, float* ms) {
    PoseidonHasher::HashOut pih = {GoldilocksField{public_inputs_hash[0]}, GoldilocksField{public_inputs_hash[1]},
                                   GoldilocksField{public_inputs_hash[2]}, GoldilocksField{public_inputs_hash[3]}};
    quotient_stack();
    Timer t(ms);
    compute_quotient_values_kernel<<<blocks32(300000), 32, 0, 0>>>(log_len, rate_bits, gf(d_points), gf(d_outs), pih, gf(d_const_sigma_leaves), 88,
                                                                  gf(d_zs_pp_leaves), 20, gf(d_wires_leaves), 234, 8, 80, 2, 231, 8, 9,
                                                                  gf(d_z_h_evals), gf(d_z_h_inverses), gf(d_k_is), gf(d_alphas), gf(d_betas),
                                                                  gf(d_gammas));
    return t.stop();
}

}  // extern "C"
