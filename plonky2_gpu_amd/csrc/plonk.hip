// plonk.hip — the permutation-argument stage of the prover on the device:
//   (1) partial products and Z      wires_permutation_partial_products_and_zs (plonky2/src/plonk/prover.rs:729-786,
//                                    plonky2/src/util/partial_products.rs:13-37)
//   (2) quotient values             compute_quotient_polys (prover.rs:790-1034) over eval_vanishing_poly_base_batch
//                                    (plonky2/src/plonk/vanishing_poly.rs:100-226) with the gate-constraint terms
//                                    supplied per point by the caller (the circuit-specific part)
// The reference keeps (1) on the host even in its GPU prover (prover.rs:322-326) — it forces a D2H/H2D
// round trip of the witness — and hard-wires (2) to one circuit (cuda/plonky2_gpu_impl.cuh:485-878).
//
// (1) is data-parallel over rows except for Z, which is a prefix product over the rows: each
// thread forms its row's chunk quotients (one Fermat inversion per chunk; the reference batches
// the inversions per row, same values), the running product is a three-kernel block scan.
// (2) is one thread per LDE point reading its leaf rows; terms are reduced with powers of alpha
// exactly in the reference's order (plonk_common.rs:97-114).
#include "plonk.h"

#include "gl_field.h"

namespace plonky2_hip {

namespace {

constexpr int MAX_CHALLENGES = 4;
constexpr int MAX_TERMS = 160;  // num_challenges * (1 + chunks) permutation terms kept per thread

struct Challenges {
    uint64_t beta[MAX_CHALLENGES], gamma[MAX_CHALLENGES], alpha[MAX_CHALLENGES];
};

// w_{2^log}^i through the two-level table of w_{2^24}
__device__ __forceinline__ uint64_t root_pow(const uint64_t *twl, const uint64_t *twh, uint32_t log, uint64_t i) {
    uint32_t e = (uint32_t)(i << (24 - log)) & 0xFFFFFFu;
    uint64_t h = twh[e >> 12];
    uint32_t lo = e & 4095u;
    return lo ? gl::mul(h, twl[lo]) : h;
}

__device__ __forceinline__ uint64_t inverse(uint64_t x) { return gl::pow(x, gl::P - 2); }

// x^(p-2): with e_k = x^(2^k - 1), p - 2 = (2^31 - 1) 2^33 + (2^32 - 1)
__device__ __forceinline__ uint64_t inverse_chain(uint64_t x) {
    auto sqn = [](uint64_t v, int k) {
        for (int i = 0; i < k; i++) v = gl::sqr(v);
        return v;
    };
    const uint64_t e2 = gl::mul(gl::sqr(x), x), e3 = gl::mul(gl::sqr(e2), x), e6 = gl::mul(sqn(e3, 3), e3), e12 = gl::mul(sqn(e6, 6), e6);
    const uint64_t e15 = gl::mul(sqn(e12, 3), e3), e30 = gl::mul(sqn(e15, 15), e15), e31 = gl::mul(gl::sqr(e30), x), e32 = gl::mul(gl::sqr(e31), x);
    return gl::mul(sqn(e31, 33), e32);
}


// One thread per (row i, challenge c): cumulative chunk quotients c_k(i) = prod_{m<=k} q_m(i) written to
// the partial-product slots (k < num_prods) and the row total r_i to the Z slot.
__global__ __launch_bounds__(256) void perm_quotients_kernel(const uint64_t *__restrict__ wires, uint64_t wires_stride,
                                                             const uint64_t *__restrict__ sigmas, uint64_t sigmas_stride,
                                                             const uint64_t *__restrict__ k_is, Challenges ch,
                                                             uint32_t num_challenges, uint32_t num_routed, uint32_t degree,
                                                             uint32_t num_prods, uint32_t log_n, const uint64_t *twl,
                                                             const uint64_t *twh, uint64_t *__restrict__ out) {
    const uint64_t n = 1ull << log_n;
    uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n * num_challenges) return;
    uint64_t i = g & (n - 1);
    uint32_t c = (uint32_t)(g >> log_n);
    const uint64_t beta = ch.beta[c], gamma = ch.gamma[c];
    const uint64_t bx = gl::mul(beta, root_pow(twl, twh, log_n, i));  // beta * x, x = w_n^i (prover_data.subgroup)
    uint64_t *z_slot = out + (uint64_t)c * n;
    uint64_t *pp_base = out + ((uint64_t)num_challenges + (uint64_t)c * num_prods) * n;
    // The chunk quotients need 1 / den_k for every chunk: one inversion for all of them (the reference's batch_multiplicative_inverse,
    // util/partial_products.rs:29-37): forward, the running products N_k of the numerators and D_k of the denominators; 1 / D_last by an
    // addition chain; backward, 1 / D_k = (1 / D_{k+1}) den_{k+1}. Ten Fermat inversions per thread were four fifths of this kernel.
    constexpr uint32_t MAX_CHUNKS = 16;
    const uint32_t chunks = (num_routed + degree - 1) / degree;
    auto chunk = [&](uint32_t k, uint64_t &num, uint64_t &den) {
        num = den = 1;
        const uint32_t j0 = k * degree, j1 = j0 + degree < num_routed ? j0 + degree : num_routed;
        for (uint32_t j = j0; j < j1; j++) {
            uint64_t w = wires[j * wires_stride + i];
            uint64_t wg = gl::add(w, gamma);
            num = gl::mul(num, gl::add(wg, gl::mul(bx, k_is[j])));                      // w + beta*k_j*x + gamma
            den = gl::mul(den, gl::add(wg, gl::mul(beta, sigmas[j * sigmas_stride + i])));  // w + beta*sigma + gamma
        }
    };
    auto store = [&](uint32_t k, uint64_t cum) {
        if (k < num_prods)
            pp_base[(uint64_t)k * n + i] = gl::canon(cum);
        else
            z_slot[i] = gl::canon(cum);
    };
    uint64_t running_n[MAX_CHUNKS], dens[MAX_CHUNKS];
    uint64_t pn = 1, pd = 1;
    if (chunks <= MAX_CHUNKS) {
        for (uint32_t k = 0; k < chunks; k++) {
            uint64_t num, den;
            chunk(k, num, den);
            pn = gl::mul(pn, num);
            pd = gl::mul(pd, den);
            running_n[k] = pn;
            dens[k] = den;
        }
    }
    if (chunks <= MAX_CHUNKS && gl::canon(pd) != 0) {
        uint64_t inv = inverse_chain(pd);  // 1 / D_k, k = chunks - 1 downwards
        for (uint32_t k = chunks; k-- > 0;) {
            store(k, gl::mul(running_n[k], inv));
            inv = gl::mul(inv, dens[k]);
        }
        return;
    }
    // a zero denominator (the reference panics there; here its chunk's quotient is 0 as before), or more chunks than the arrays hold
    uint64_t cum = 1;
    for (uint32_t k = 0; k < chunks; k++) {
        uint64_t num, den;
        chunk(k, num, den);
        cum = gl::mul(cum, gl::mul(num, inverse(den)));
        store(k, cum);
    }
}

// ---- exclusive prefix product over rows (three kernels) ---------------------------------------
constexpr int SCAN_T = 256, SCAN_E = 4, SCAN_B = SCAN_T * SCAN_E;

__device__ __forceinline__ uint64_t block_exclusive_scan(uint64_t v, uint64_t *lds, uint64_t *total) {
    const uint32_t t = threadIdx.x;
    lds[t] = v;
    __syncthreads();
    for (int off = 1; off < SCAN_T; off <<= 1) {
        uint64_t x = (t >= (uint32_t)off) ? gl::mul(lds[t - off], lds[t]) : lds[t];
        __syncthreads();
        lds[t] = x;
        __syncthreads();
    }
    uint64_t incl = lds[t];
    uint64_t excl = t ? lds[t - 1] : 1;
    if (total && t == SCAN_T - 1) *total = incl;
    __syncthreads();
    return excl;
}

// in place: v[i] <- product of the block's earlier elements; totals[blk] <- product of the block
__global__ __launch_bounds__(SCAN_T) void scan_blocks_kernel(uint64_t *v, uint64_t n, uint64_t col_stride, uint64_t *totals,
                                                             uint64_t totals_stride) {
    __shared__ uint64_t lds[SCAN_T];
    uint64_t *col = v + (uint64_t)blockIdx.y * col_stride;
    uint64_t base = (uint64_t)blockIdx.x * SCAN_B + (uint64_t)threadIdx.x * SCAN_E;
    uint64_t e[SCAN_E];
    uint64_t p = 1;
#pragma unroll
    for (int k = 0; k < SCAN_E; k++) {
        e[k] = base + k < n ? col[base + k] : 1;
        p = gl::mul(p, e[k]);
    }
    uint64_t excl = block_exclusive_scan(p, lds, totals + (uint64_t)blockIdx.y * totals_stride + blockIdx.x);
#pragma unroll
    for (int k = 0; k < SCAN_E; k++) {
        if (base + k < n) col[base + k] = gl::canon(excl);
        excl = gl::mul(excl, e[k]);
    }
}

// exclusive scan of m block totals per column by one workgroup
__global__ __launch_bounds__(SCAN_T) void scan_totals_kernel(uint64_t *totals, uint64_t m, uint64_t totals_stride) {
    __shared__ uint64_t lds[SCAN_T];
    uint64_t *t = totals + (uint64_t)blockIdx.x * totals_stride;
    uint64_t per = (m + SCAN_T - 1) / SCAN_T;
    uint64_t lo = (uint64_t)threadIdx.x * per, hi = lo + per < m ? lo + per : m;
    uint64_t p = 1;
    for (uint64_t i = lo; i < hi; i++) p = gl::mul(p, t[i]);
    uint64_t excl = block_exclusive_scan(p, lds, nullptr);
    for (uint64_t i = lo; i < hi; i++) {
        uint64_t x = t[i];
        t[i] = excl;
        excl = gl::mul(excl, x);
    }
}

// Z[i] = block prefix * local exclusive; pp_k[i] = Z[i] * c_k(i)
__global__ __launch_bounds__(256) void perm_finalize_kernel(uint64_t *out, const uint64_t *totals, uint64_t totals_stride,
                                                            uint32_t num_challenges, uint32_t num_prods, uint32_t log_n) {
    const uint64_t n = 1ull << log_n;
    uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n * num_challenges) return;
    uint64_t i = g & (n - 1);
    uint32_t c = (uint32_t)(g >> log_n);
    uint64_t z = gl::mul(out[(uint64_t)c * n + i], totals[(uint64_t)c * totals_stride + i / SCAN_B]);
    out[(uint64_t)c * n + i] = gl::canon(z);
    uint64_t *pp_base = out + ((uint64_t)num_challenges + (uint64_t)c * num_prods) * n;
    for (uint32_t k = 0; k < num_prods; k++) pp_base[(uint64_t)k * n + i] = gl::canon(gl::mul(z, pp_base[(uint64_t)k * n + i]));
}

// ---- quotient values ---------------------------------------------------------------------------
// Table-driven gate constraints: evaluate_gate_constraints_base_batch (plonky2/src/plonk/vanishing_poly.rs:267-306)
// over Gate::eval_filtered (gates/gate.rs:86-109) and compute_filter (gates/gate.rs:261-268). Every gate of
// the circuit is a small register program (see plonky2_gpu_amd/gate_program.py for the instruction set);
// all lanes of a wavefront execute the same instruction stream, so the interpreter does not diverge.
constexpr int GP_MAX_REGS = 64, GP_MAX_CONSTRAINTS = 256;
enum : uint16_t { GP_LOAD_WIRE, GP_LOAD_CONST, GP_LOAD_PI, GP_LOAD_IMM, GP_ADD, GP_SUB, GP_MUL, GP_EMIT, GP_MULK, GP_ACC, GP_ACCR };

struct GateProgramDev {
    const uint16_t *instrs;  // 4 x u16 per instruction: op, dst, a, b
    const uint32_t *gates;   // 6 x u32 per gate: row, selector_index, group_start, group_end, prog_start, prog_len
    const uint64_t *imms;
    uint32_t num_gates, num_selectors, num_gate_constraints;
    uint64_t pih[4];
};

__device__ void eval_gate_program(const GateProgramDev &gp, const uint64_t *local_constants, uint64_t c_es,
                                  const uint64_t *local_wires, uint64_t w_es, uint64_t *acc) {
    uint64_t regs[GP_MAX_REGS];
    for (uint32_t k = 0; k < gp.num_gate_constraints; k++) acc[k] = 0;
    for (uint32_t g = 0; g < gp.num_gates; g++) {
        const uint32_t *d = gp.gates + 6 * g;
        const uint32_t row = d[0], si = d[1], gs = d[2], ge = d[3], ps = d[4], pl = d[5];
        const uint64_t s = local_constants[si * c_es];
        uint64_t filt = 1;
        for (uint32_t i = gs; i < ge; i++)
            if (i != row) filt = gl::mul(filt, gl::sub(i, s));
        if (gp.num_selectors > 1) filt = gl::mul(filt, gl::sub(0xFFFFFFFFull, s));  // UNUSED_SELECTOR, selectors.rs:11
        uint32_t k = 0;
        uint64_t acc_lo[4] = {0, 0, 0, 0}, acc_hi[4] = {0, 0, 0, 0};  // GP_ACC: plain (wrapping-free by contract) column sums
        for (uint32_t pc = ps; pc < ps + pl; pc++) {
            const uint16_t *in = gp.instrs + 4 * pc;
            const uint16_t op = in[0], dst = in[1] & (GP_MAX_REGS - 1), a = in[2], b = in[3];
            switch (op) {
                case GP_LOAD_WIRE: regs[dst] = local_wires[a * w_es]; break;
                case GP_LOAD_CONST: regs[dst] = local_constants[(gp.num_selectors + a) * c_es]; break;
                case GP_LOAD_PI: regs[dst] = gp.pih[a & 3]; break;
                case GP_LOAD_IMM: regs[dst] = gp.imms[a]; break;
                case GP_ADD: regs[dst] = gl::add(regs[a & (GP_MAX_REGS - 1)], regs[b & (GP_MAX_REGS - 1)]); break;
                case GP_SUB: regs[dst] = gl::sub(regs[a & (GP_MAX_REGS - 1)], regs[b & (GP_MAX_REGS - 1)]); break;
                case GP_MUL: regs[dst] = gl::mul(regs[a & (GP_MAX_REGS - 1)], regs[b & (GP_MAX_REGS - 1)]); break;
                case GP_MULK: regs[dst] = gl::mul(regs[a & (GP_MAX_REGS - 1)], 1ull << (b & 63)); break;  // b < 64 here
                case GP_ACC: {  // acc[dst] += r[a] * imm[b], imm < 2^32: the two 32-bit halves of r[a] in separate u64 sums
                    const uint64_t x = regs[a & (GP_MAX_REGS - 1)], c = gp.imms[b];
                    acc_lo[dst & 3] += (x & 0xFFFFFFFFull) * c;
                    acc_hi[dst & 3] += (x >> 32) * c;
                    break;
                }
                case GP_ACCR:  // r[dst] = acc[a] mod p; acc[a] = 0
                    regs[dst] = gl::fold96(acc_lo[a & 3], acc_hi[a & 3]);
                    acc_lo[a & 3] = acc_hi[a & 3] = 0;
                    break;
                case GP_EMIT:
                    if (k < gp.num_gate_constraints) acc[k] = gl::add(acc[k], gl::mul(filt, regs[a & (GP_MAX_REGS - 1)]));
                    k++;
                    break;
                default: break;
            }
        }
    }
}

struct QuotientParams {
    GateProgramDev gp;
    uint32_t has_program;
    const uint64_t *wires_leaves, *cs_leaves, *zpp_leaves, *k_is, *gate_terms, *gate_partial, *twl, *twh;
    uint64_t w_rs, w_es, c_rs, c_es, z_rs, z_es;  // element j of leaf t at base[t*rs + j*es]
    uint64_t *out;  // [num_challenges][lde_size]
    uint32_t wires_len, cs_len, zpp_len, num_constants, num_routed, num_challenges, degree, num_prods;
    uint32_t degree_bits, rate_bits, qdb, num_gate_constraints;
    uint64_t shift, g_pow_n;
    Challenges ch;
};

// One thread per LEAF t < lde_size. get_lde_values(i, step) reads leaf reverse_bits(i*step, bits)
// (fri/oracle.rs:1007-1018), and reverse_bits(i << step_log, log_lde + step_log) == reverse_bits(i, log_lde),
// so the points of the quotient domain are exactly the first lde_size leaves and leaf t holds point
// i = reverse_bits(t, log_lde): consecutive threads read consecutive leaves — contiguous in every column
// of the column-major LDE — and only the 8-byte result is scattered.
__global__ __launch_bounds__(128) void quotient_values_kernel(const QuotientParams p) {
    const uint32_t log_lde = p.degree_bits + p.qdb;
    const uint64_t lde_size = 1ull << log_lde, n = 1ull << p.degree_bits;
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= lde_size) return;
    const uint64_t i = log_lde ? (__brevll(t) >> (64 - log_lde)) : 0;
    const uint64_t i_next = (i + (1ull << p.qdb)) & (lde_size - 1);  // the point g*x
    const uint64_t t_next = log_lde ? (__brevll(i_next) >> (64 - log_lde)) : 0;
    const uint64_t *wires = p.wires_leaves + t * p.w_rs;
    const uint64_t *cs = p.cs_leaves + t * p.c_rs;
    const uint64_t *sig = cs + p.num_constants * p.c_es;
    const uint64_t *zpp = p.zpp_leaves + t * p.z_rs;
    const uint64_t *zpp_next = p.zpp_leaves + t_next * p.z_rs;

    const uint64_t x = gl::mul(p.shift, root_pow(p.twl, p.twh, log_lde, i));  // shifted_x (prover.rs:903)
    // Z_H(x) = g^n * v^(i mod rate) - 1 (field/src/zero_poly_coset.rs:20-41)
    const uint64_t v = p.qdb ? root_pow(p.twl, p.twh, p.qdb, i & ((1ull << p.qdb) - 1)) : 1;
    const uint64_t zh = gl::sub(gl::mul(p.g_pow_n, v), 1);
    const uint64_t zh_inv = inverse(zh);
    const uint64_t l0 = gl::mul(zh, inverse(gl::mul(n, gl::sub(x, 1))));  // eval_l_0 (zero_poly_coset.rs:57-60)

    uint64_t terms[MAX_TERMS];
    uint32_t nt = 0;
    for (uint32_t c = 0; c < p.num_challenges; c++) terms[nt++] = gl::mul(l0, gl::sub(zpp[c * p.z_es], 1));
    for (uint32_t c = 0; c < p.num_challenges; c++) {
        const uint64_t beta = p.ch.beta[c], gamma = p.ch.gamma[c];
        const uint64_t bx = gl::mul(beta, x);
        uint64_t prev = zpp[c * p.z_es];
        uint32_t k = 0;
        for (uint32_t j0 = 0; j0 < p.num_routed; j0 += p.degree, k++) {
            uint64_t num = 1, den = 1;
            uint32_t j1 = j0 + p.degree < p.num_routed ? j0 + p.degree : p.num_routed;
            for (uint32_t j = j0; j < j1; j++) {
                uint64_t wg = gl::add(wires[j * p.w_es], gamma);
                num = gl::mul(num, gl::add(wg, gl::mul(bx, p.k_is[j])));
                den = gl::mul(den, gl::add(wg, gl::mul(beta, sig[j * p.c_es])));
            }
            uint64_t next = (k < p.num_prods) ? zpp[(p.num_challenges + c * p.num_prods + k) * p.z_es] : zpp_next[c * p.z_es];
            // check_partial_products (util/partial_products.rs:52-76): prev*num - next*den
            terms[nt++] = gl::sub(gl::mul(prev, num), gl::mul(next, den));
            prev = next;
        }
    }
    // reduce_with_powers_multi (plonk_common.rs:97-114): Horner from the LAST term over
    // [L_0 (Z-1)] | [partial-product checks] | [gate constraints]
    const uint64_t *gt = p.gate_terms ? p.gate_terms + i * p.num_gate_constraints : nullptr;
    uint64_t gate_acc[GP_MAX_CONSTRAINTS];
    if (p.has_program) {
        eval_gate_program(p.gp, cs, p.c_es, wires, p.w_es, gate_acc);
        gt = gate_acc;
    }
    for (uint32_t c = 0; c < p.num_challenges; c++) {
        const uint64_t alpha = p.ch.alpha[c];
        // the gate-constraint tail of the Horner sum, already reduced by the compiled gate kernel
        uint64_t cumul = p.gate_partial ? p.gate_partial[(uint64_t)c * lde_size + t] : 0;
        if (gt)
            for (uint32_t q = p.num_gate_constraints; q-- > 0;) cumul = gl::mac(gt[q], cumul, alpha);
        for (uint32_t q = nt; q-- > 0;) cumul = gl::mac(terms[q], cumul, alpha);
        p.out[(uint64_t)c * lde_size + i] = gl::canon(gl::mul(cumul, zh_inv));  // prover.rs:985-991
    }
}

unsigned grid_for(uint64_t n, unsigned block) { return (unsigned)((n + block - 1) / block); }

}  // namespace

uint32_t num_partial_products(uint32_t num_routed, uint32_t degree) { return (num_routed + degree - 1) / degree - 1; }

hipError_t permutation_partial_products(const NttTables &tb, const uint64_t *wires, uint64_t wires_stride, const uint64_t *sigmas,
                                        uint64_t sigmas_stride, const uint64_t *k_is, const uint64_t *betas, const uint64_t *gammas,
                                        uint32_t num_challenges, uint32_t num_routed, uint32_t degree, uint32_t log_n, uint64_t *out,
                                        hipStream_t stream) {
    if (num_challenges == 0 || num_challenges > MAX_CHALLENGES || degree < 2 || num_routed == 0 || log_n > 24)
        return hipErrorInvalidValue;
    const uint64_t n = 1ull << log_n;
    const uint32_t num_prods = num_partial_products(num_routed, degree);
    Challenges ch = {};
    for (uint32_t c = 0; c < num_challenges; c++) {
        ch.beta[c] = betas[c] % glh::P;
        ch.gamma[c] = gammas[c] % glh::P;
    }
    hipLaunchKernelGGL(perm_quotients_kernel, dim3(grid_for(n * num_challenges, 256)), dim3(256), 0, stream, wires, wires_stride,
                       sigmas, sigmas_stride, k_is, ch, num_challenges, num_routed, degree, num_prods, log_n, tb.twl, tb.twh, out);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    // Z(x_i) = prod_{t<i} r_t: exclusive prefix product of the Z slots (Z(1) = 1)
    const uint64_t blocks = (n + SCAN_B - 1) / SCAN_B;
    if (!tb.scratch || tb.scratch_elems < blocks * num_challenges) return hipErrorInvalidValue;
    uint64_t *totals = tb.scratch;
    hipLaunchKernelGGL(scan_blocks_kernel, dim3((unsigned)blocks, num_challenges), dim3(SCAN_T), 0, stream, out, n, n, totals, blocks);
    hipLaunchKernelGGL(scan_totals_kernel, dim3(num_challenges), dim3(SCAN_T), 0, stream, totals, blocks, blocks);
    hipLaunchKernelGGL(perm_finalize_kernel, dim3(grid_for(n * num_challenges, 256)), dim3(256), 0, stream, out, totals, blocks,
                       num_challenges, num_prods, log_n);
    return hipGetLastError();
}

namespace {
// ---- the same kernel for the case the prover runs: gate constraints from the run-time compiled kernels (gate_partial) or none ------
// quotient_values_kernel above serves every source of gate constraints, the interpreter included, and pays for it on every path: 3.8 KB
// of scratch per lane (the interpreter's registers, its constraint sums, the term array), two Fermat inversions per point
// (127 multiplications each) — 4.4 ms of a 8.8 ms quotient at n = 2^18 (profiles/r05_quotient_kernel_trace.txt). Here:
//   * no arrays: the permutation terms are produced last to first and go straight into the Horner sums of all challenges
//     (reduce_with_powers_multi, plonk_common.rs:97-114, runs from the last term);
//   * Z_H(x) takes 2^qdb values on the quotient domain (zero_poly_coset.rs:20-41): they and their inverses come from the host;
//   * the one inversion left, 1 / (n (x - 1)) for L_0 (zero_poly_coset.rs:57-60), by an addition chain for p - 2 = 2^64 - 2^32 - 1:
//     64 squarings and 9 multiplications.
struct ZhTable {
    uint64_t zh[16], zh_inv[16];
};

template <int NCH>
__global__ __launch_bounds__(128) void quotient_values_fast_kernel(const QuotientParams p, const ZhTable zt) {
    const uint32_t log_lde = p.degree_bits + p.qdb;
    const uint64_t lde_size = 1ull << log_lde, n = 1ull << p.degree_bits;
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= lde_size) return;
    const uint64_t i = log_lde ? (__brevll(t) >> (64 - log_lde)) : 0;
    const uint64_t i_next = (i + (1ull << p.qdb)) & (lde_size - 1);  // the point g*x
    const uint64_t t_next = log_lde ? (__brevll(i_next) >> (64 - log_lde)) : 0;
    const uint64_t *wires = p.wires_leaves + t * p.w_rs;
    const uint64_t *sig = p.cs_leaves + t * p.c_rs + p.num_constants * p.c_es;
    const uint64_t *zpp = p.zpp_leaves + t * p.z_rs;
    const uint64_t *zpp_next = p.zpp_leaves + t_next * p.z_rs;
    const uint64_t x = gl::mul(p.shift, root_pow(p.twl, p.twh, log_lde, i));  // shifted_x (prover.rs:903)
    uint64_t zh = zt.zh[0], zh_inv = zt.zh_inv[0];
    const uint32_t which = (uint32_t)i & ((1u << p.qdb) - 1);
    for (uint32_t e = 1; e < (1u << p.qdb); e++) {
        zh = which == e ? zt.zh[e] : zh;
        zh_inv = which == e ? zt.zh_inv[e] : zh_inv;
    }
    // the gate-constraint tail of the Horner sums, already reduced by the compiled gate kernels
    uint64_t cumul[NCH];
#pragma unroll
    for (int c = 0; c < NCH; c++) cumul[c] = p.gate_partial ? p.gate_partial[(uint64_t)c * lde_size + t] : 0;
    const uint32_t chunks = p.num_prods + 1;
#pragma unroll
    for (int c = NCH - 1; c >= 0; c--) {
        const uint64_t beta = p.ch.beta[c], gamma = p.ch.gamma[c];
        const uint64_t bx = gl::mul(beta, x);
        for (uint32_t k = chunks; k-- > 0;) {
            const uint32_t j0 = k * p.degree, j1 = j0 + p.degree < p.num_routed ? j0 + p.degree : p.num_routed;
            uint64_t num = 1, den = 1;
            for (uint32_t j = j0; j < j1; j++) {
                const uint64_t wg = gl::add(wires[j * p.w_es], gamma);
                num = gl::mul(num, gl::add(wg, gl::mul(bx, p.k_is[j])));
                den = gl::mul(den, gl::add(wg, gl::mul(beta, sig[j * p.c_es])));
            }
            const uint64_t prev = k == 0 ? zpp[c * p.z_es] : zpp[(NCH + c * p.num_prods + k - 1) * p.z_es];
            const uint64_t next = k < p.num_prods ? zpp[(NCH + c * p.num_prods + k) * p.z_es] : zpp_next[c * p.z_es];
            // check_partial_products (util/partial_products.rs:52-76): prev*num - next*den
            const uint64_t term = gl::sub(gl::mul(prev, num), gl::mul(next, den));
#pragma unroll
            for (int cc = 0; cc < NCH; cc++) cumul[cc] = gl::mac(term, cumul[cc], p.ch.alpha[cc]);
        }
    }
    const uint64_t l0 = gl::mul(zh, inverse_chain(gl::mul(n, gl::sub(x, 1))));  // eval_l_0 (zero_poly_coset.rs:57-60)
#pragma unroll
    for (int c = NCH - 1; c >= 0; c--) {
        const uint64_t term = gl::mul(l0, gl::sub(zpp[c * p.z_es], 1));
#pragma unroll
        for (int cc = 0; cc < NCH; cc++) cumul[cc] = gl::mac(term, cumul[cc], p.ch.alpha[cc]);
    }
#pragma unroll
    for (int c = 0; c < NCH; c++) p.out[(uint64_t)c * lde_size + i] = gl::canon(gl::mul(cumul[c], zh_inv));  // prover.rs:985-991
}
}  // namespace

hipError_t quotient_values(const NttTables &tb, const QuotientArgs &a, uint64_t *out, hipStream_t stream) {
    if (a.num_challenges == 0 || a.num_challenges > MAX_CHALLENGES || a.quotient_degree_factor < 2 || a.num_routed == 0)
        return hipErrorInvalidValue;
    uint32_t qdb = 0;
    while ((1u << qdb) < a.quotient_degree_factor) qdb++;  // log2_ceil
    if (qdb > a.rate_bits || a.degree_bits + qdb > 24) return hipErrorInvalidValue;
    const uint32_t num_prods = num_partial_products(a.num_routed, a.quotient_degree_factor);
    if (a.num_challenges * (2 + num_prods) > MAX_TERMS) return hipErrorInvalidValue;
    if (a.zpp_len < a.num_challenges * (1 + num_prods) || a.cs_len < a.num_constants + a.num_routed || a.wires_len < a.num_routed)
        return hipErrorInvalidValue;
    QuotientParams p = {};
    p.wires_leaves = a.wires_leaves;
    p.cs_leaves = a.cs_leaves;
    p.zpp_leaves = a.zpp_leaves;
    p.k_is = a.k_is;
    p.gate_terms = a.gate_terms;
    p.twl = tb.twl;
    p.twh = tb.twh;
    p.out = out;
    p.wires_len = a.wires_len;
    p.cs_len = a.cs_len;
    p.zpp_len = a.zpp_len;
    if (a.column_stride) {  // the column-major LDE [leaf_len][column_stride]
        if (a.column_stride < (1ull << (a.degree_bits + qdb))) return hipErrorInvalidValue;
        p.w_rs = p.c_rs = p.z_rs = 1;
        p.w_es = p.c_es = p.z_es = a.column_stride;
    } else {  // leaf-major rows [n_ext][leaf_len]
        p.w_rs = a.wires_len, p.c_rs = a.cs_len, p.z_rs = a.zpp_len;
        p.w_es = p.c_es = p.z_es = 1;
    }
    p.num_constants = a.num_constants;
    p.num_routed = a.num_routed;
    p.num_challenges = a.num_challenges;
    p.degree = a.quotient_degree_factor;
    p.num_prods = num_prods;
    p.degree_bits = a.degree_bits;
    p.rate_bits = a.rate_bits;
    p.qdb = qdb;
    p.num_gate_constraints = (a.gate_terms || a.gate_program) ? a.num_gate_constraints : 0;
    if (a.gate_kernel) {
        if (a.gate_terms || a.gate_program || !a.gate_partial_workspace || !a.public_inputs_hash) return hipErrorInvalidValue;
        if (gate_kernel_num_challenges(a.gate_kernel) != a.num_challenges) return hipErrorInvalidValue;
        // the compiled programs index wire and constant columns by literal: they must exist in the leaves handed over
        if (gate_kernel_wires_needed(a.gate_kernel) > a.wires_len || gate_kernel_constants_needed(a.gate_kernel) > a.num_constants)
            return hipErrorInvalidValue;
        hipError_t ge = gate_kernel_launch(a.gate_kernel, a.wires_leaves, p.w_rs, p.w_es, a.cs_leaves, p.c_rs, p.c_es, a.alphas,
                                           a.public_inputs_hash, 1ull << (a.degree_bits + qdb), a.gate_partial_workspace, stream);
        if (ge != hipSuccess) return ge;
        p.gate_partial = a.gate_partial_workspace;
    }
    if (a.gate_program) {
        if (a.gate_terms || a.num_gate_constraints > GP_MAX_CONSTRAINTS) return hipErrorInvalidValue;
        const GateProgramArgs &g = *a.gate_program;
        if (g.num_selectors > a.num_constants) return hipErrorInvalidValue;
        p.has_program = 1;
        p.gp.instrs = g.instrs;
        p.gp.gates = g.gates;
        p.gp.imms = g.imms;
        p.gp.num_gates = g.num_gates;
        p.gp.num_selectors = g.num_selectors;
        p.gp.num_gate_constraints = a.num_gate_constraints;
        for (int k = 0; k < 4; k++) p.gp.pih[k] = g.public_inputs_hash[k] % glh::P;
    }
    p.shift = a.shift % glh::P;
    p.g_pow_n = glh::pow(p.shift, 1ull << a.degree_bits);
    for (uint32_t c = 0; c < a.num_challenges; c++) {
        p.ch.beta[c] = a.betas[c] % glh::P;
        p.ch.gamma[c] = a.gammas[c] % glh::P;
        p.ch.alpha[c] = a.alphas[c] % glh::P;
    }
    const uint64_t lde_size = 1ull << (a.degree_bits + qdb);
    if (!p.has_program && !p.gate_terms && qdb <= 4) {
        ZhTable zt = {};
        const uint64_t w = glh::root_of_unity(qdb);
        for (uint32_t e = 0; e < (1u << qdb); e++) {  // Z_H(x) = g^n * w^(i mod 2^qdb) - 1 (zero_poly_coset.rs:20-41)
            zt.zh[e] = glh::add(glh::mul(p.g_pow_n, glh::pow(w, e)), glh::P - 1);
            zt.zh_inv[e] = glh::inv(zt.zh[e]);
        }
        const dim3 grid(grid_for(lde_size, 128)), block(128);
        switch (a.num_challenges) {
            case 1: hipLaunchKernelGGL(quotient_values_fast_kernel<1>, grid, block, 0, stream, p, zt); break;
            case 2: hipLaunchKernelGGL(quotient_values_fast_kernel<2>, grid, block, 0, stream, p, zt); break;
            case 3: hipLaunchKernelGGL(quotient_values_fast_kernel<3>, grid, block, 0, stream, p, zt); break;
            default: hipLaunchKernelGGL(quotient_values_fast_kernel<4>, grid, block, 0, stream, p, zt); break;
        }
        return hipGetLastError();
    }
    hipLaunchKernelGGL(quotient_values_kernel, dim3(grid_for(lde_size, 128)), dim3(128), 0, stream, p);
    return hipGetLastError();
}

}  // namespace plonky2_hip

// ---- evaluation of base-field polynomials at points of the quadratic extension ------------------
// OpeningSet::new (plonky2/src/plonk/proof.rs:305-334): every committed polynomial is evaluated at
// zeta (and the Z polynomials at g*zeta) in F_{p^2} = F_p[X]/(X^2 - 7) (field/src/goldilocks_extensions.rs:13-26,
// extension/quadratic.rs:173-185), p.to_extension().eval(z) = Horner (field/src/polynomial/mod.rs:161-166).
// The reference copies all coefficients back to the host for this (fri/oracle.rs:403-407, 462); here they
// are read once where they already live. Grid = (segments, polynomials): a workgroup owns one contiguous
// segment of one polynomial, thread t does Horner with z^256 over coefficients t, t+256, ...; the
// per-thread values are weighted by z^t, summed in LDS, weighted by z^(segment start) and written as one
// partial per (point, polynomial, segment); a second tiny kernel adds the segments.
namespace plonky2_hip {
namespace {

struct Ext2 {
    uint64_t a, b;  // a + b*X, X^2 = 7
};

__device__ __forceinline__ Ext2 ext_mul(Ext2 x, Ext2 y) {
    // c0 = a0*b0 + 7*a1*b1, c1 = a0*b1 + a1*b0 (quadratic.rs:176-184)
    uint64_t t = gl::mul(x.b, y.b);
    uint64_t t7 = gl::sub(gl::mul_pow2<3>(t), t);
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

constexpr int EV_T = 256, EV_MAX_POINTS = 4;

struct EvalPoints {
    uint64_t z0[EV_MAX_POINTS], z1[EV_MAX_POINTS];
    uint32_t count;
};

__global__ __launch_bounds__(EV_T) void eval_ext2_kernel(const uint64_t *__restrict__ coeffs, uint64_t stride, uint32_t log_n,
                                                         uint32_t log_seg, EvalPoints pts, uint64_t *__restrict__ partials) {
    __shared__ uint64_t red[2][EV_T];
    const uint32_t t = threadIdx.x, seg = blockIdx.x, poly = blockIdx.y;
    const uint64_t seg_len = 1ull << log_seg, seg_start = (uint64_t)seg << log_seg;
    const uint64_t *c = coeffs + (uint64_t)poly * stride + seg_start;
    const uint32_t n_seg = 1u << (log_n - log_seg);
    for (uint32_t q = 0; q < pts.count; q++) {
        const Ext2 z{pts.z0[q], pts.z1[q]};
        const Ext2 zs = ext_pow(z, EV_T);  // z^256
        const uint64_t zs1_7 = gl::sub(gl::mul_pow2<3>(zs.b), zs.b);
        Ext2 acc{0, 0};
        // coefficients t + 256k, highest k first: acc = acc * z^256 + c
        for (int64_t k = (int64_t)(seg_len / EV_T) - 1; k >= 0; k--) {
            uint64_t cv = (t + (uint64_t)k * EV_T < seg_len) ? c[t + (uint64_t)k * EV_T] : 0;
            uint64_t n0 = gl::mac(gl::mac(cv, acc.a, zs.a), acc.b, zs1_7);
            uint64_t n1 = gl::mac(gl::mul(acc.a, zs.b), acc.b, zs.a);
            acc = Ext2{n0, n1};
        }
        if (seg_len < EV_T) {  // tiny polynomials: one coefficient (or none) per thread
            acc = Ext2{t < seg_len ? c[t] : 0, 0};
        }
        Ext2 w = ext_mul(acc, ext_pow(z, t));
        red[0][t] = w.a;
        red[1][t] = w.b;
        __syncthreads();
        for (int off = EV_T / 2; off > 0; off >>= 1) {
            if (t < (uint32_t)off) {
                red[0][t] = gl::add(red[0][t], red[0][t + off]);
                red[1][t] = gl::add(red[1][t], red[1][t + off]);
            }
            __syncthreads();
        }
        if (t == 0) {
            Ext2 r = ext_mul(Ext2{red[0][0], red[1][0]}, ext_pow(z, seg_start));
            uint64_t *o = partials + (((uint64_t)q * gridDim.y + poly) * n_seg + seg) * 2;
            o[0] = gl::canon(r.a);
            o[1] = gl::canon(r.b);
        }
        __syncthreads();
    }
}

__global__ void eval_ext2_sum_kernel(const uint64_t *partials, uint32_t n_seg, uint64_t total, uint64_t *out) {
    uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= total) return;
    uint64_t a = 0, b = 0;
    for (uint32_t s = 0; s < n_seg; s++) {
        a = gl::add(a, partials[(g * n_seg + s) * 2]);
        b = gl::add(b, partials[(g * n_seg + s) * 2 + 1]);
    }
    out[g * 2] = gl::canon(a);
    out[g * 2 + 1] = gl::canon(b);
}

}  // namespace

hipError_t eval_polys_ext2(const NttTables &tb, const uint64_t *coeffs, uint64_t n_polys, uint32_t log_n, uint64_t stride,
                           const uint64_t *points, uint32_t n_points, uint64_t *out, hipStream_t stream) {
    if (n_points == 0 || n_points > EV_MAX_POINTS || log_n > 32 || n_polys > 65535) return hipErrorInvalidValue;
    if (n_polys == 0) return hipSuccess;
    // enough segments to give the chip >= ~2048 workgroups, each at least 4096 coefficients long
    uint32_t log_seg = log_n;
    while (log_seg > 12 && (n_polys << (log_n - log_seg)) < 2048) log_seg--;
    const uint32_t n_seg = 1u << (log_n - log_seg);
    const uint64_t need = (uint64_t)n_points * n_polys * n_seg * 2;
    if (!tb.scratch || tb.scratch_elems < need) return hipErrorInvalidValue;
    EvalPoints pts = {};
    pts.count = n_points;
    for (uint32_t q = 0; q < n_points; q++) {
        pts.z0[q] = points[2 * q] % glh::P;
        pts.z1[q] = points[2 * q + 1] % glh::P;
    }
    hipLaunchKernelGGL(eval_ext2_kernel, dim3(n_seg, (unsigned)n_polys), dim3(EV_T), 0, stream, coeffs, stride, log_n, log_seg, pts,
                       tb.scratch);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const uint64_t total = (uint64_t)n_points * n_polys;
    hipLaunchKernelGGL(eval_ext2_sum_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, tb.scratch, n_seg, total, out);
    return hipGetLastError();
}

}  // namespace plonky2_hip
