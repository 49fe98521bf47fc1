/* plonky2_hip.h — C ABI of libplonky2_hip.so: the MI355X (gfx950) replacement for the reference's
 * `cuda/` crate (sideprotocol/plonky2-gpu), covering the prover's data-parallel hot path only:
 * Goldilocks NTT / inverse NTT / coset LDE, Poseidon Merkle-cap construction and the
 * PolynomialBatch commit; the permutation argument, the quotient polynomials for any circuit (gates as
 * register programs), openings and the FRI opening pipeline; and, on top of those, a whole prove() in one call.
 * No circuit building, witness generation or verification.
 *
 * Two groups of entry points:
 *   (A) the reference's own extern "C" symbols (cuda/src/lib.rs:58-145 <-> cuda/plonky2_gpu.cu),
 *       same names, argument order and memory-layout contract, so the Rust side links unchanged;
 *   (B) a generic-shape `gl_*` API with 64-bit sizes, explicit strides and real error returns,
 *       which (A) is implemented on and which new host code should prefer.
 *
 * Conventions
 *   - Field elements are plain-domain Goldilocks u64 (p = 2^64 - 2^32 + 1). Inputs may be any
 *     u64 representative (field/src/goldilocks_field.rs:26); every OUTPUT buffer holds canonical
 *     values (< p), i.e. exactly what the reference yields after `to_canonical_u64`.
 *   - All `d_*` pointers are DEVICE pointers. The callee allocates nothing for data: the caller
 *     owns every buffer (as in the reference, plonky2/src/fri/oracle.rs:94-106). The library keeps,
 *     per device, a few hundred KiB of read-only twiddle tables, and per CONTEXT one workspace of
 *     gl_workspace_bytes() = 512 MiB (allocated by gl_ctx_create(), or on the first call that sees a
 *     caller-built context; gl_ctx_set_workspace() hands in the caller's own memory instead, for a
 *     host that sizes all device memory up front) that the natural-order transforms, the scans
 *     (partial products, divide_by_linear), the opening partial sums, gl_sponge_absorb,
 *     gl_merkle_open_batch and gl_fri_proof_of_work stage through, plus the commit's event pair and
 *     its low-priority hashing stream.
 *     Several contexts on one device run CONCURRENTLY (up to version 0.5 they took turns): nothing
 *     mutable is shared between them, so two host threads, each with its own context, keep two
 *     proofs in flight on one GPU and each fills the other's latency-bound phases (transcript,
 *     small tree layers, openings). One context is used by one host thread at a time; data shared
 *     by two contexts is the caller's to order, as with any two streams. Two things are still taken
 *     in turns, because their constant tables exist once: launches of ONE compiled gate kernel
 *     (gl_gate_kernel_build; proofs of one circuit handle on two contexts overlap everywhere
 *     except in the gate kernels) and the reference symbol compute_quotient_polys per device.
 *     A circuit handle (gl_circuit_create) may be proven with from several contexts at once; it
 *     keeps one buffer pool per context.
 *     Three entry points do allocate device memory themselves: gl_circuit_create (the preprocessed
 *     commitment, freed by gl_circuit_destroy) and gl_prove (every buffer of one proof), because their
 *     job is to own a whole computation, and the reference symbol compute_quotient_polys (a staging
 *     buffer, unless the caller provides it: gl_reference_quotient_set_staging).
 *   - Errors are returned BY VALUE as {code, message}; code 0 = success; `message` is
 *     malloc'ed (strdup) and owned by the caller, who frees it with free() — the convention of
 *     cuda/src/lib.rs:21-35 / cuda/plonky2_gpu.cu:19-31.
 *   - `ctx` points to {hipStream_t stream; hipStream_t stream2;} — the HIP twin of the reference's
 *     CudaInnerContext (plonky2/src/fri/oracle.rs:43-47, cuda/plonky2_gpu.cu:4-7). Create one with
 *     gl_ctx_create() or hand in your own pair of streams.
 *   - (B) calls are ASYNCHRONOUS on ctx->stream unless stated otherwise; (A) calls synchronise
 *     the stream before returning, like the reference (cuda/plonky2_gpu.cu:82).
 */
#ifndef PLONKY2_HIP_H
#define PLONKY2_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* The library is built with -fvisibility=hidden: exactly the functions declared in this header are exported. */
#pragma GCC visibility push(default)

/* == cuda::Error (cuda/src/lib.rs:21-25) / RustError (cuda/plonky2_gpu.cu:19-31) */
typedef struct GlError {
    int code;      /* 0 = ok; >0 = hipError_t; <0 = library error (GL_E_*) */
    char *message; /* NULL or strdup'ed, caller frees */
} GlError;

#define GL_E_INVALID (-1)     /* bad argument (sizes, alignment, NULL) */
#define GL_E_UNSUPPORTED (-2) /* entry point outside the hot path of this build */

/* == DataSlice (cuda/src/lib.rs:52-56): host struct holding a device pointer + i32 length */
typedef struct GlDataSlice {
    const void *ptr;
    int len;
} GlDataSlice;

/* ---------------------------------------------------------------------------------------------
 * (B) generic API
 * ------------------------------------------------------------------------------------------- */

/* Context = two HIP streams on one device (layout-compatible with CudaInnerContext). */
int gl_device_count(void);
void *gl_ctx_create(int device); /* NULL on failure; also makes `device` current */
void gl_ctx_destroy(void *ctx);
GlError gl_ctx_synchronize(void *ctx); /* waits for both streams */
/* A caller-built context ({stream, stream2} made with the host's own HIP binding, e.g. the reference's
 * CudaInnerContext through rustacuda_hip) gets its library-side state (workspace, events, hashing stream) on the
 * first call that sees it, keyed by ctx->stream; gl_ctx_release() waits for both streams and gives that state back
 * (before the caller destroys its streams). gl_ctx_destroy() = gl_ctx_release() + destroying the streams it created. */
void gl_ctx_release(void *ctx);
/* The workspace of a context: gl_workspace_bytes() bytes (constant for a build: 512 MiB). gl_ctx_set_workspace()
 * replaces the library's allocation by `d_workspace` (>= gl_workspace_bytes() bytes, 16-byte aligned, on the context's
 * device, the caller's to free after gl_ctx_release / gl_ctx_destroy) — the reference's memory contract, in which the
 * caller sizes every device buffer once up front (fri/oracle.rs:94-106) and the callee allocates nothing.
 * d_workspace = NULL goes back to a library-owned one. Waits for the context's stream. */
uint64_t gl_workspace_bytes(void);
GlError gl_ctx_set_workspace(void *ctx, void *d_workspace, uint64_t bytes);

/* Thin device-memory helpers for hosts without a HIP binding (synchronous). gl_malloc allocates on the calling
 * thread's CURRENT device; gl_ctx_malloc on the device of `ctx` (and makes it current) — what a host with several
 * contexts, or with threads that never called hipSetDevice, should use. Every entry point that takes a ctx runs on
 * the ctx's device in the same way, whatever the thread's current device was. */
GlError gl_malloc(void **d_ptr, uint64_t bytes);
GlError gl_ctx_malloc(void **d_ptr, uint64_t bytes, void *ctx);
GlError gl_free(void *d_ptr);
/* Page-locked host staging memory — the reference's MyAllocator (plonky2/src/fri/oracle.rs:49-73,
 * cudaHostAlloc): transfers from it run at link speed and do not stage through a bounce buffer. */
GlError gl_malloc_host(void **h_ptr, uint64_t bytes);
GlError gl_free_host(void *h_ptr);
GlError gl_memcpy_h2d(void *d_dst, const void *h_src, uint64_t bytes, void *ctx);
/* The same copy queued on ctx->stream2 WITHOUT waiting for it: with h_src page-locked (gl_malloc_host) it runs on
 * the DMA engines while ctx->stream computes — upload the next proof's witness during gl_prove of the current one
 * (gl_prove works on ctx->stream and waits for both streams only when it is done), then gl_ctx_synchronize()
 * before using d_dst. The witness upload (468 MiB, 8.5 ms at the ed25519 shape) disappears from the proof period. */
GlError gl_memcpy_h2d_async(void *d_dst, const void *h_src, uint64_t bytes, void *ctx);
GlError gl_memcpy_d2h(void *h_dst, const void *d_src, uint64_t bytes, void *ctx);
GlError gl_memcpy_d2d(void *d_dst, const void *d_src, uint64_t bytes, void *ctx);
GlError gl_memset_zero(void *d_dst, uint64_t bytes, void *ctx);

/* HIP-event timing on ctx->stream (what bench.py brackets kernels with). */
GlError gl_event_create(void **event);
GlError gl_event_record(void *event, void *ctx);
GlError gl_event_elapsed_ms(float *ms, void *start_event, void *stop_event); /* syncs on stop */
void gl_event_destroy(void *event);

/* Batched NTT, in place. Polynomial i occupies d_values[i*stride .. i*stride + 2^log_n).
 *   inverse = 0: fft_with_options(.., zero_factor None)  (field/src/fft.rs:58-66)
 *   inverse = 1: ifft_with_options                        (field/src/fft.rs:73-103)
 *   bit_reversed = 1 (forward only): output slot m holds the value of natural index bitrev(m),
 *     i.e. reverse_index_bits(fft(x)) (util/src/lib.rs:188) — the Merkle leaf order.
 * log_n <= 24. For inverse, stride must be a multiple of 2^log_n. */
GlError gl_ntt_batch(uint64_t *d_values, uint64_t poly_num, uint32_t log_n, uint64_t stride, int inverse,
                     int bit_reversed, void *ctx);

/* Coset low-degree extension of poly_num coefficient vectors (length 2^log_n) to
 * 2^(log_n+rate_bits) evaluations on shift*H, BIT-REVERSED order:
 *   d_out[i*dst_stride + m] = coset_fft_with_options(lde(coeffs_i, rate_bits), shift)[bitrev(m)]
 * (field/src/polynomial/mod.rs:205-207, 286-299; fri/oracle.rs:979-1004 then :942-952 per column).
 * d_out must not overlap d_coeffs. */
GlError gl_coset_lde_batch(const uint64_t *d_coeffs, uint64_t *d_out, uint64_t poly_num, uint32_t log_n,
                           uint32_t rate_bits, uint64_t shift, uint64_t src_stride, uint64_t dst_stride, void *ctx);

/* Natural-order coset transforms, in place:
 *   inverse = 0: PolynomialCoeffs::coset_fft(shift)   (field/src/polynomial/mod.rs:281-299)
 *   inverse = 1: PolynomialValues::coset_ifft(shift)  (field/src/polynomial/mod.rs:64-77) — the
 *               last step of compute_quotient_polys (plonk/prover.rs:1009-1021). */
GlError gl_coset_ntt_batch(uint64_t *d_values, uint64_t poly_num, uint32_t log_n, uint64_t stride, uint64_t shift, int inverse,
                           void *ctx);

/* Partial products and Z of the permutation argument for every challenge
 * (wires_permutation_partial_products_and_zs, plonky2/src/plonk/prover.rs:702-786;
 * quotient_chunk_products / partial_products_and_z_gx, plonky2/src/util/partial_products.rs:13-37).
 *   d_wires [>= num_routed][2^log_n] column-major, column stride wires_stride (the witness layout)
 *   d_sigmas [num_routed][2^log_n] column-major (prover_data.sigmas transposed), d_k_is [num_routed]
 *   h_betas / h_gammas: HOST arrays of num_challenges elements
 *   d_out [num_challenges * (1 + num_prods)][2^log_n], num_prods = ceil(num_routed / qdf) - 1, in the
 *   order the prover commits them: every Z first, then the partial products challenge-major
 *   (prover.rs:112-117) — ready for gl_commit_from_values. */
GlError gl_permutation_partial_products(const uint64_t *d_wires, uint64_t wires_stride, const uint64_t *d_sigmas,
                                        uint64_t sigmas_stride, const uint64_t *d_k_is, const uint64_t *h_betas,
                                        const uint64_t *h_gammas, uint32_t num_challenges, uint32_t num_routed,
                                        uint32_t quotient_degree_factor, uint32_t log_n, uint64_t *d_out, void *ctx);

/* compute_quotient_polys (plonky2/src/plonk/prover.rs:790-1034) for any circuit: the permutation
 * terms, L_0(x)(Z(x)-1), the alpha reduction and the division by Z_H are evaluated here
 * (plonk/vanishing_poly.rs:100-226, plonk_common.rs:97-114, field/src/zero_poly_coset.rs); the
 * circuit-specific gate-constraint terms are an INPUT, one row of num_gate_constraints values per
 * LDE point in natural point order (NULL = no gate constraints). The three *_leaves pointers are the
 * leaf-major LDE rows of the commitments (d_leaves of gl_commit_*), read at leaf
 * reverse_bits(i * step) like PolynomialBatch::get_lde_values (fri/oracle.rs:1007-1018).
 * d_quotient_polys [num_challenges][n << log2_ceil(qdf)] receives the COEFFICIENTS (after coset_ifft,
 * prover.rs:1009-1021); chunking into degree-n pieces is a reinterpretation (prover.rs:153-166). */
/* Table-driven gate constraints for gl_compute_quotient_polys: evaluate_gate_constraints_base_batch
 * (plonky2/src/plonk/vanishing_poly.rs:267-306) with compute_filter (gates/gate.rs:261-268) for ANY circuit.
 * Each gate is a register program of GlGateInstr {op, dst, a, b} (u16 each):
 *   0 LOAD_WIRE dst<-local_wires[a]      1 LOAD_CONST dst<-local_constants[num_selectors+a]
 *   2 LOAD_PI dst<-public_inputs_hash[a]  3 LOAD_IMM dst<-d_immediates[a]
 *   4 ADD  5 SUB  6 MUL  dst<-r[a] op r[b]       7 EMIT next constraint of the gate <- r[a]
 *   8 MULK dst<-r[a] * 2^b  (b < 64; a shift in the run-time compiled kernel)
 *   9 ACC  acc[dst] += r[a] * d_immediates[b]   (4 accumulators, dst = 0..3; the immediate must be < 2^32)
 *  10 ACCR dst<-acc[a] mod p; acc[a]<-0
 *     Sums with small constant weights without a modular step per term — limb recombinations sum limb_j * B^j,
 *     MDS rows — the way plonky2 computes its MDS layer on x86 (hash/poseidon.rs:34-47 uses the same split): an
 *     accumulator is a pair of plain u64 sums over the low and the high 32-bit halves of the operands, one
 *     32x32+64 multiply-add each per term, folded as lo + hi*2^32 mod p by ACCR. CONTRACT: between two ACCRs of an
 *     accumulator, sum of immediates * (2^32 - 1) < 2^63, so that neither half can wrap. gl_gate_kernel_build checks
 *     this and refuses the program otherwise; the interpreter cannot (its program is in device memory), the
 *     emitters in plonky2_gpu_amd/gate_program.py enforce it when they generate code. Accumulators start at 0 in
 *     every gate.
 * (64 registers). Gate g is described by GlGateDesc {row = its index in the circuit's gate list,
 * selector_index, group_start, group_end (selectors_info.groups[selector_index]), prog_start, prog_len}.
 * Constraint k of every gate accumulates into term k, multiplied by the gate's filter. */
typedef struct GlGateInstr {
    uint16_t op, dst, a, b;
} GlGateInstr;
typedef struct GlGateDesc {
    uint32_t row, selector_index, group_start, group_end, prog_start, prog_len;
} GlGateDesc;
typedef struct GlGateProgram {
    const GlGateInstr *d_instrs;  /* device */
    const GlGateDesc *d_gates;    /* device */
    const uint64_t *d_immediates; /* device, may be NULL */
    uint32_t num_gates, num_selectors;
    uint64_t public_inputs_hash[4];
} GlGateProgram;

/* Emitters of the register programs for the gate kinds of the ed25519 gate list, so that a compiled host needs no Python to
 * describe its circuit: what each gate's eval_unfiltered_base_one computes (plonky2/src/gates/{noop,constant,public_input,
 * arithmetic_base,base_sum,random_access,poseidon}.rs, u32/src/gates/{add_many_u32,arithmetic_u32,subtraction_u32,
 * range_check_u32,comparison}.rs), as a program in the encoding above — with the ACC / ACCR accumulators wherever a sum has
 * small constant weights, and their overflow contract enforced at emission. params by kind:
 *   NOOP, PUBLIC_INPUT, POSEIDON: none        CONSTANT {num_consts}         ARITHMETIC {num_ops}
 *   BASE_SUM {B, num_limbs}                   U32_ADD_MANY {num_addends, num_ops}
 *   U32_ARITHMETIC / U32_SUBTRACTION {num_ops}  U32_RANGE_CHECK {num_input_limbs}
 *   COMPARISON {num_bits, num_chunks} (chunks of at most 4 bits)   RANDOM_ACCESS {bits, num_copies, num_extra_constants}
 *   ARITHMETIC_EXTENSION / MUL_EXTENSION {num_ops}   REDUCING / REDUCING_EXTENSION {num_coeffs}   EXPONENTIATION {num_power_bits}
 *   POSEIDON_MDS: none   LOW_DEGREE_INTERPOLATION / HIGH_DEGREE_INTERPOLATION {subgroup_bits <= 4}
 *   (plonky2/src/gates/{arithmetic_extension,multiplication_extension,reducing,reducing_extension,exponentiation,poseidon_mds,
 *   low_degree_interpolation,high_degree_interpolation}.rs; pairs of wires are elements of F_p[X]/(X^2 - 7), plonk/vars.rs:122-129)
 * gl_gate_programs_emit builds the programs of a whole gate list in circuit order (gate i has selector index
 * gates[i].selector_index; group_bounds[2 s], group_bounds[2 s + 1] = selectors_info.groups[s], gates/selectors.rs): host arrays
 * in exactly the form GlCircuitDesc / gl_gate_kernel_build take, immediates deduplicated across the list, num_gate_constraints =
 * the largest number of constraints of any gate. The arrays are malloc'd by the library: release them with
 * gl_gate_programs_free. Pure host code: no device is needed. */
enum GlGateKind {
    GL_GATE_NOOP = 0, GL_GATE_CONSTANT = 1, GL_GATE_PUBLIC_INPUT = 2, GL_GATE_ARITHMETIC = 3, GL_GATE_BASE_SUM = 4,
    GL_GATE_U32_ADD_MANY = 5, GL_GATE_U32_ARITHMETIC = 6, GL_GATE_U32_SUBTRACTION = 7, GL_GATE_U32_RANGE_CHECK = 8,
    GL_GATE_COMPARISON = 9, GL_GATE_RANDOM_ACCESS = 10, GL_GATE_POSEIDON = 11,
    /* the other gates of upstream plonky2 (what standard_recursion_config circuits are made of), extension degree D = 2 */
    GL_GATE_ARITHMETIC_EXTENSION = 12, GL_GATE_MUL_EXTENSION = 13, GL_GATE_REDUCING = 14, GL_GATE_REDUCING_EXTENSION = 15,
    GL_GATE_EXPONENTIATION = 16, GL_GATE_POSEIDON_MDS = 17, GL_GATE_LOW_DEGREE_INTERPOLATION = 18, GL_GATE_HIGH_DEGREE_INTERPOLATION = 19
};
typedef struct GlGateSpec {
    uint32_t kind;      /* GlGateKind */
    uint32_t params[3];
    uint32_t selector_index;
} GlGateSpec;
typedef struct GlGatePrograms {
    GlGateInstr *instrs;
    GlGateDesc *gates;
    uint64_t *immediates;
    uint32_t num_instrs, num_gates, num_immediates, num_gate_constraints;
} GlGatePrograms;
GlError gl_gate_programs_emit(const GlGateSpec *gates, uint32_t num_gates, const uint32_t *group_bounds, uint32_t num_selectors,
                              GlGatePrograms *out);
void gl_gate_programs_free(GlGatePrograms *programs);

/* The same gate programs compiled at run time (hiprtc, gfx950) into a kernel specialised to the circuit:
 * gates that share wires and subexpressions are generated together (fused units: each value is loaded / computed once and goes to every
 * gate's accumulators), registers in VGPRs, immediates as literals or scalar operands. Built once per circuit
 * (under a second for a few small gates; the 25-gate ed25519 list: about a minute of CPU, in six units — 15 s when forked hiprtc workers
 * compile them side by side, PLONKY2_HIP_JIT_FORK=1), reused for every proof; h_* are HOST arrays in the GlGateInstr / GlGateDesc encoding.
 * On failure the GlError message carries the compiler log. */
GlError gl_gate_kernel_build(const GlGateInstr *h_instrs, uint32_t num_instrs, const GlGateDesc *h_gates, uint32_t num_gates,
                             const uint64_t *h_immediates, uint32_t num_immediates, uint32_t num_selectors,
                             uint32_t num_gate_constraints, uint32_t num_challenges, void **kernel);
void gl_gate_kernel_destroy(void *kernel);
/* the generated HIP source (owned by the kernel object) — for inspection and tests */
const char *gl_gate_kernel_source(const void *kernel);

typedef struct GlQuotientArgs {
    const uint64_t *d_wires_leaves;
    const uint64_t *d_constants_sigmas_leaves;
    const uint64_t *d_zs_partial_products_leaves;
    uint32_t wires_leaf_len, constants_sigmas_leaf_len, zs_partial_products_leaf_len;
    const uint64_t *d_k_is;
    const uint64_t *d_gate_constraint_terms; /* may be NULL */
    const uint64_t *h_betas, *h_gammas, *h_alphas; /* HOST, num_challenges each */
    uint32_t num_constants, num_routed_wires, num_challenges, num_gate_constraints;
    uint32_t degree_bits, rate_bits, quotient_degree_factor;
    uint64_t coset_shift; /* F::coset_shift() = 7 */
    const GlGateProgram *gate_program; /* HOST struct, may be NULL; exclusive with d_gate_constraint_terms */
    /* 0: the three d_*_leaves are leaf-major rows [n_ext][leaf_len] (d_leaves of gl_commit_*);
     * otherwise they are the column-major LDE [leaf_len][column_stride] in bit-reversed row order
     * (d_lde of gl_commit_*) — coalesced reads, and no leaf-major copy has to exist at all. */
    uint64_t column_stride;
    /* Gates compiled by gl_gate_kernel_build (may be NULL; exclusive with the two other sources of gate
     * constraints). Needs h_public_inputs_hash (4, host) and d_gate_workspace
     * [num_challenges][n << log2_ceil(qdf)] (device scratch). */
    const void *gate_kernel;
    const uint64_t *h_public_inputs_hash;
    uint64_t *d_gate_workspace;
} GlQuotientArgs;
GlError gl_compute_quotient_polys(const GlQuotientArgs *args, uint64_t *d_quotient_polys, void *ctx);

/* Evaluate poly_num base-field polynomials (coefficients [poly_num][2^log_n], column stride `stride`)
 * at num_points (<= 4) points of the quadratic extension F_p[X]/(X^2 - 7): what OpeningSet::new does
 * with every commitment's `polynomials` at zeta and g*zeta (plonky2/src/plonk/proof.rs:305-334;
 * p.to_extension().eval(z), field/src/polynomial/mod.rs:161-166; extension/quadratic.rs:173-185).
 * h_points: HOST array of num_points pairs (c0, c1) meaning c0 + c1*X.
 * d_out[(q*poly_num + i)*2 + {0,1}] = polynomial i at point q, canonical. */
GlError gl_eval_polys_ext2(const uint64_t *d_coeffs, uint64_t poly_num, uint32_t log_n, uint64_t stride, const uint64_t *h_points,
                           uint32_t num_points, uint64_t *d_out, void *ctx);

/* ---- FRI opening pipeline primitives (PolynomialBatch::prove_openings, plonky2/src/fri/oracle.rs:1047-1112;
 * fri_proof, plonky2/src/fri/prover.rs). Extension-field vectors (F_p[X]/(X^2-7)) are PLANAR on the
 * device: v[0..len) first components, v[len..2len) second components, so that every transform of an
 * extension polynomial is two columns of gl_ntt_batch / gl_coset_lde_batch. Host-side scalars
 * (h_alpha, h_point, h_scale, h_beta) are pairs (c0, c1).
 *
 * gl_fri_reduce_polys_base: d_out = sum_j alpha^j * poly_j (ReducingFactor::reduce_polys_base,
 *   util/reducing.rs:83-95); d_poly_ptrs is a DEVICE array of num_polys device pointers to base
 *   polynomials of n coefficients each.
 * gl_fri_divide_by_linear: q = (p(X) - p(z))/(X - z) (polynomial/division.rs:75-88) of the composition
 *   polynomial (destroyed), written shifted by one: final[0] = 0, final[i+1] = (accumulate ?
 *   final[i+1]*scale : 0) + q_i  — the update `alpha.shift_poly(&mut final_poly); final_poly += quotient`
 *   plus the multiplication by X of oracle.rs:1069-1087.
 * gl_fri_fold: out[k] = sum_{i < 2^arity_bits} c[k*2^arity_bits + i] * beta^i (prover.rs:103-111).
 * gl_ext2_interleave: rows[2i + c] = plane_c[i] (flatten(), prover.rs:90-95): consecutive groups of
 *   `arity` extension values become the Merkle leaves of a commit-phase tree (gl_merkle_tree_from_leaves
 *   with leaf_len = 2*arity).
 * gl_fri_proof_of_work (SYNCHRONOUS): h_state = the challenger's sponge state with its buffered inputs
 *   already written, witness_pos = input_buffer.len(); returns the SMALLEST witness w such that
 *   permute(state with state[witness_pos] = w)[7] has >= min_leading_zeros leading zero bits
 *   (prover.rs:122-171; the reference's rayon find_any returns an arbitrary one). */
GlError gl_fri_reduce_polys_base(const uint64_t *const *d_poly_ptrs, uint32_t num_polys, uint64_t n, const uint64_t *h_alpha,
                                 uint64_t *d_out, void *ctx);
GlError gl_fri_divide_by_linear(uint64_t *d_composition, uint64_t n, const uint64_t *h_point, const uint64_t *h_scale, int accumulate,
                                uint64_t *d_final, void *ctx);
GlError gl_fri_fold(const uint64_t *d_coeffs, uint64_t len, uint32_t arity_bits, const uint64_t *h_beta, uint64_t *d_out, void *ctx);
/* The same with beta read from device memory (two canonical words, e.g. where gl_challenger_step wrote them): the FRI commit phase
 * (fri/prover.rs:77-120) runs without a host synchronisation per layer. */
GlError gl_fri_fold_device(const uint64_t *d_coeffs, uint64_t len, uint32_t arity_bits, const uint64_t *d_beta, uint64_t *d_out, void *ctx);
GlError gl_ext2_interleave(const uint64_t *d_planes, uint64_t len, uint64_t *d_rows, void *ctx);
GlError gl_fri_proof_of_work(const uint64_t *h_state, uint32_t witness_pos, uint32_t min_leading_zeros, uint64_t *h_witness, void *ctx);
/* The same against a device-resident Challenger: the duplex state is the challenger's (its input buffer written over the state's
 * first words, the candidate behind it, fri/prover.rs:137-147); the witness lands in *d_witness (from where gl_challenger_step can
 * observe it) and in *h_witness. Synchronous like gl_fri_proof_of_work. */
GlError gl_fri_proof_of_work_device(const uint64_t *d_challenger, uint32_t min_leading_zeros, uint64_t *d_witness, uint64_t *h_witness, void *ctx);

/* count Poseidon permutations in place, states[count][12] (plonky2/src/hash/poseidon.rs:602-616). */
GlError gl_poseidon_permute_batch(uint64_t *d_states, uint64_t count, void *ctx);

/* The transcript's sponge: h_state[12] (in/out, host) absorbs n_blocks full rate-8 blocks of h_inputs in
 * overwrite mode, one permutation per block — Challenger::duplexing (plonky2/src/iop/challenger.rs:131-149)
 * repeated, and hash_n_to_hash_no_pad (hash/hashing.rs:81-108) for whole blocks. SYNCHRONOUS. */
GlError gl_sponge_absorb(uint64_t *h_state, const uint64_t *h_inputs, uint32_t n_blocks, void *ctx);

/* The Challenger with its state in DEVICE memory (plonky2/src/iop/challenger.rs:19-149): d_challenger = 32 u64 the caller owns
 * (sponge state, input buffer, the two buffer lengths). One call = one launch, asynchronous on ctx->stream, no host round trip:
 * observe_elements over up to eight device sources in order — read where the producing kernels left them: a cap straight from
 * gl_commit_* / gl_merkle_tree_*, openings from gl_eval_polys_ext2, a planar extension vector (planar_len != 0: element i is
 * d_ptr[(i & 1) * planar_len + (i >> 1)], i.e. observe_extension_elements) — then get_n_challenges(n_challenges) into d_out
 * (canonical). GL_CHALLENGER_RESET starts from the empty transcript (Challenger::new) before observing. GL_CHALLENGER_HASH:
 * instead of challenges, d_out[0..4) = hash_n_to_hash_no_pad of everything observed since the reset (hash/hashing.rs:81-108; use it
 * with GL_CHALLENGER_RESET on a scratch challenger). Values a host holds (circuit digest, public inputs) are observed from a device
 * copy. The host fetches the challenges it needs itself with one gl_memcpy_d2h per step; kernels that can read a challenge from
 * device memory need no fetch at all (gl_fri_fold_device, gl_merkle_open_batch_device, gl_fri_proof_of_work_device). */
typedef struct GlObserveSrc {
    const uint64_t *d_ptr;
    uint64_t count;      /* field elements observed from this source */
    uint64_t planar_len; /* 0: d_ptr[i]; else the plane length of an extension vector kept as [2][planar_len] */
} GlObserveSrc;
#define GL_CHALLENGER_RESET 1u
#define GL_CHALLENGER_HASH 2u
GlError gl_challenger_step(uint64_t *d_challenger, const GlObserveSrc *h_srcs, uint32_t n_srcs, uint32_t n_challenges, uint64_t *d_out,
                           uint32_t flags, void *ctx);

/* MerkleTree::prove (plonky2/src/hash/merkle_tree.rs:392-440) and the leaf itself for `count` leaf
 * indices in one launch and one copy — what fri_prover_query_round (fri/prover.rs:199-260) does per
 * query and tree. Element j of leaf i is read at d_leaves[i*row_stride + j*elem_stride] (leaf-major rows:
 * (leaf_len, 1); the LDE's columns: (1, n_leaves)). Host outputs: h_out_leaves[count][leaf_len],
 * h_out_siblings[count][log2(n_leaves) - cap_height][4]. SYNCHRONOUS. */
GlError gl_merkle_open_batch(const uint64_t *d_leaves, uint64_t row_stride, uint64_t elem_stride, uint32_t leaf_len, uint64_t n_leaves,
                             uint32_t cap_height, const uint64_t *d_digests, const uint64_t *h_indices, uint32_t count,
                             uint64_t *h_out_leaves, uint64_t *h_out_siblings, void *ctx);
/* The same with everything in device memory and nothing synchronised: leaf of query q = (d_indices[q] mod (n_leaves << index_shift))
 * >> index_shift — d_indices may be raw challenges (fri/prover.rs:186-190: x_index = challenge mod lde size) and index_shift the
 * sum of the FRI arities before this layer's tree (:213-236). Outputs [count][leaf_len] and [count][layers][4] in device memory. */
GlError gl_merkle_open_batch_device(const uint64_t *d_leaves, uint64_t row_stride, uint64_t elem_stride, uint32_t leaf_len, uint64_t n_leaves,
                                    uint32_t cap_height, const uint64_t *d_digests, const uint64_t *d_indices, uint32_t count,
                                    uint32_t index_shift, uint64_t *d_out_leaves, uint64_t *d_out_siblings, void *ctx);

/* MerkleTree::new (plonky2/src/hash/merkle_tree.rs:283-319) over n_leaves (power of two) leaves of
 * leaf_len elements; leaf hash = hash_or_noop (plonk/config.rs:56-67).
 *   _columns: d_cols[j*col_stride + i] = element j of leaf i   (the NTT's output layout)
 *   _leaves : d_rows[i*leaf_len + j]                            (the reference's Vec<Vec<F>>)
 * d_digests: 4*2*(n_leaves - 2^cap_height) u64, reference layout (merkle_tree.rs:46-54);
 * d_cap: 4*2^cap_height u64. cap_height > log2(n_leaves) -> GL_E_INVALID (the reference panics). */
GlError gl_merkle_tree_from_columns(const uint64_t *d_cols, uint32_t leaf_len, uint64_t n_leaves, uint64_t col_stride,
                                    uint32_t cap_height, uint64_t *d_digests, uint64_t *d_cap, void *ctx);
GlError gl_merkle_tree_from_leaves(const uint64_t *d_rows, uint32_t leaf_len, uint64_t n_leaves, uint32_t cap_height,
                                   uint64_t *d_digests, uint64_t *d_cap, void *ctx);

/* d_cols[c*col_stride + r] -> d_rows[r*n_cols + c]  (plonky2/src/util/mod.rs:23-53) */
GlError gl_transpose(const uint64_t *d_cols, uint64_t *d_rows, uint32_t n_cols, uint64_t n_rows, uint64_t col_stride,
                     void *ctx);

/* The pack step of a commitment whose columns are sharded over `world` GPUs (one process per GPU; plonky2_gpu_amd/dist.py):
 * d_out[(q*n_cols + c)*leaves_per_rank + i] = d_lde[c*col_stride + q*leaves_per_rank + i] — for every rank q, the leaf
 * range it will hash of every one of this rank's n_cols LDE columns, contiguous per destination rank, in ONE launch
 * (d_out: world * n_cols * leaves_per_rank elements). The slices go out as they are with one send per peer. */
GlError gl_pack_leaf_ranges(const uint64_t *d_lde, uint64_t col_stride, uint32_t n_cols, uint64_t leaves_per_rank, uint32_t world,
                            uint64_t *d_out, void *ctx);

/* PolynomialBatch::from_coeffs (plonky2/src/fri/oracle.rs:911-977) without the host-side struct:
 *   d_coeffs   [poly_num][2^log_n]            in
 *   d_lde      [(poly_num+salt_size)][n_ext]  out, column-major, bit-reversed (n_ext = 2^(log_n+rate_bits));
 *              the salt_size trailing columns are read as given (caller-provided randomness,
 *              oracle.rs:998-1002) and take part in the leaf hash. They are read by kernels on a stream of the library's own
 *              that starts behind everything queued on ctx's stream at the time of the call: write them on ctx's stream
 *              (gl_memcpy_*, a kernel launched there) or complete the writes before calling.
 *   d_leaves   [n_ext][poly_num+salt_size]    out, leaf-major (= merkle_tree.leaves); may be NULL. MAY overlap d_coeffs (the
 *              reference's caller passes one region for both, fri/oracle.rs:409-422): the coefficients are then consumed --
 *              the leaves replace them. d_lde must not overlap d_coeffs or d_leaves.
 *   d_digests / d_cap as gl_merkle_tree_*.
 * shift is F::coset_shift() = 7 in the reference (field/src/types.rs:431-433). */
GlError gl_commit_from_coeffs(const uint64_t *d_coeffs, uint64_t poly_num, uint32_t log_n, uint32_t rate_bits,
                              uint32_t cap_height, uint32_t salt_size, uint64_t shift, uint64_t *d_lde,
                              uint64_t *d_leaves, uint64_t *d_digests, uint64_t *d_cap, void *ctx);
/* PolynomialBatch::from_values (oracle.rs:709-731): d_values is transformed IN PLACE into the
 * coefficients (= PolynomialBatch.polynomials), then as gl_commit_from_coeffs. */
GlError gl_commit_from_values(uint64_t *d_values, uint64_t poly_num, uint32_t log_n, uint32_t rate_bits,
                              uint32_t cap_height, uint32_t salt_size, uint64_t shift, uint64_t *d_lde,
                              uint64_t *d_leaves, uint64_t *d_digests, uint64_t *d_cap, void *ctx);

/* ---- the whole prover in two calls ---------------------------------------------------------------
 * gl_circuit_create = the prover-side part of CircuitBuilder::build (plonky2/src/plonk/circuit_builder.rs:
 * 849-960): uploads sigmas / k_is, commits constants||sigmas (constants_sigmas_commitment), derives the
 * circuit digest (:915-927, empty domain separator) unless one is given, and compiles the gate programs.
 * gl_prove = prove() (plonky2/src/plonk/prover.rs:40-233) from the full witness on: d_wires is the flat
 * [num_wires][2^degree_bits] matrix of wire values in HBM (MatrixWitness::my_wire_values,
 * iop/witness.rs:351-362; left untouched). The proof comes back in the reference's wire format
 * (write_proof_with_public_inputs, util/serialization.rs:674-689) in a malloc'd buffer (gl_bytes_free).
 * h_stage_ms (optional, GL_PROVE_STAGES doubles) receives per-stage wall times with a device
 * synchronisation at each boundary: wires commit, partial products, Z/pp commit, quotient, quotient
 * commit, opening set, FRI combine, FRI commit phase, proof of work, query rounds, serialisation. */
typedef struct GlFriParams {
    uint32_t rate_bits, cap_height, proof_of_work_bits, num_query_rounds;
    uint32_t num_reductions;
    const uint32_t *reduction_arity_bits; /* host, num_reductions */
    uint32_t hiding; /* FriParams::hiding = CircuitConfig::zero_knowledge (fri/mod.rs:64-65, plonk/circuit_data.rs:74): the wires, Zs /
                      * partial products and quotient commitments carry SALT_SIZE = 4 random elements per leaf (fri/oracle.rs:41,
                      * 985-1002); such a circuit is proved with gl_prove_zk */
} GlFriParams;
typedef struct GlCircuitDesc {
    uint32_t struct_size; /* = sizeof(GlCircuitDesc) of the header the caller was compiled against. The struct embeds GlFriParams by
                           * value and has grown before (`hiding`, round 4): a caller built against another layout gets GL_E_INVALID
                           * from gl_circuit_create instead of shifted fields. (A caller of the 0.3 layout passes its degree_bits here,
                           * which no size equals.) */
    uint32_t degree_bits, num_wires, num_routed_wires, num_constants, num_challenges, quotient_degree_factor;
    uint32_t num_gate_constraints;
    GlFriParams fri;
    const uint64_t *h_k_is;      /* num_routed_wires */
    const uint64_t *h_constants; /* [num_constants][n] value columns, selectors first */
    const uint64_t *h_sigmas;    /* [num_routed_wires][n] value columns */
    const GlGateInstr *h_instrs;
    uint32_t num_instrs;
    const GlGateDesc *h_gates;
    uint32_t num_gates;
    const uint64_t *h_immediates;
    uint32_t num_immediates, num_selectors;
    int compile_gates;                /* 1: run-time compiled kernel, 0: interpreter */
    const uint64_t *h_circuit_digest; /* 4, or NULL to derive it */
} GlCircuitDesc;
#define GL_PROVE_STAGES 11
GlError gl_circuit_create(const GlCircuitDesc *desc, void **circuit, void *ctx);
void gl_circuit_destroy(void *circuit);
/* gl_prove recycles its working buffers from proof to proof of the same circuit (one proof's worth of
 * HBM stays attached to the circuit PER CONTEXT that proves with it: ~10 GB at n = 2^18 with 234 wires, plus a
 * page-locked staging buffer of a few hundred KiB for what travels between host and device). gl_circuit_trim
 * releases them without destroying the circuit; the next proof allocates again. Call it only between proofs.
 * Since 0.6 the proof's transcript lives in device memory (gl_challenger_step): gl_prove synchronises with the
 * device where the HOST computes with a challenge (betas / gammas, alphas, zeta, the FRI alpha), for the
 * proof-of-work witness, and once for everything that goes into the proof bytes. */
GlError gl_circuit_trim(void *circuit);
/* circuit digest (4) and constants_sigmas cap (4 << cap_height): what VerifierOnlyCircuitData holds */
GlError gl_circuit_info(const void *circuit, uint64_t *h_digest, uint64_t *h_constants_sigmas_cap);
GlError gl_prove(const void *circuit, const uint64_t *d_wires, const uint64_t *h_public_inputs, uint32_t num_public_inputs,
                 uint8_t **proof, uint64_t *proof_len, double *h_stage_ms, void *ctx);
/* prove() of a circuit built with zero_knowledge = true (fri.hiding != 0): the three blinded commitments — wires, Zs / partial
 * products, quotient chunks (plonk/prover.rs:84, 125, 174; PlonkOracle::*.blinding, plonk/plonk_common.rs:20-44) — get SALT_SIZE = 4
 * extra elements per leaf, which the proof's initial-tree openings carry and the verifier strips (fri/proof.rs:45-52).
 * The reference draws them from OsRng (F::rand_vec, fri/oracle.rs:998-1002); here the randomness is the CALLER's:
 *   d_salts  [3][4][n_ext] uniform field elements, any u64 representative (words >= p are reduced on their way into the commitment, so
 *            the proof carries canonical words like the reference's rand_vec output) (n_ext = 2^(degree_bits + rate_bits)): block 0 for the wires commitment, 1 for
 *            Zs / partial products, 2 for the quotient; column k of a block is element k of the salt of every leaf, in LEAF order
 *            (entry j belongs to leaf j, the leaf of the LDE point bitrev(j)).
 * gl_prove on a hiding circuit and gl_prove_zk on a non-hiding one return GL_E_INVALID. */
/* `count` independent proofs of ONE circuit with `in_flight` of them at a time — the per-GPU unit of a batch of proofs (a
 * throughput job: one proof alone leaves the chip idle or at one wavefront in its latency-bound phases — transcript, small tree
 * layers, openings; a second one in flight fills them: 1.15 x the proofs per second at the ed25519 shape). Worker w, a host thread
 * the call starts, proves witnesses w, w + in_flight, .. on ctxs[w] (in_flight distinct contexts of the circuit's device, each
 * with its own 512 MiB workspace); d_wires[i] / h_public_inputs[i] / proofs[i] / proof_lens[i] belong to proof i, every proof is
 * what gl_prove gives for that witness alone; all or nothing on error. Not for hiding circuits (use gl_prove_zk per proof). */
GlError gl_prove_many(const void *circuit, const uint64_t *const *d_wires, const uint64_t *const *h_public_inputs, uint32_t num_public_inputs,
                      uint32_t count, uint8_t **proofs, uint64_t *proof_lens, void *const *ctxs, uint32_t in_flight);
GlError gl_prove_zk(const void *circuit, const uint64_t *d_wires, const uint64_t *h_public_inputs, uint32_t num_public_inputs,
                    const uint64_t *d_salts, uint8_t **proof, uint64_t *proof_len, double *h_stage_ms, void *ctx);
void gl_bytes_free(uint8_t *p);

/* ---------------------------------------------------------------------------------------------
 * (A) the reference's extern "C" surface (cuda/src/lib.rs:58-145). Synchronous.
 * ------------------------------------------------------------------------------------------- */

/* lib.rs:59 — declared by the reference, its definition is commented out there
 * (cuda/plonky2_gpu.cu:59-67). Here: makes device 0 current and builds the twiddle tables. */
void init(void);

/* lib.rs:61-69 / plonky2_gpu.cu:70-86. In-place inverse NTT of poly_num polynomials of
 * values_num_per_poly = 2^log_len elements each, contiguous. root_table is accepted and ignored
 * (twiddles are internal); n_inv is a HOST pointer and is checked against 2^-log_len. */
GlError ifft(uint64_t *d_values_flatten, int poly_num, int values_num_per_poly, int log_len,
             const uint64_t *d_root_table, const uint64_t *n_inv, void *ctx);

/* lib.rs:100-114 / plonky2_gpu.cu:435-606. Region contract (element offsets from
 * d_ext_values_flatten, P = poly_num, S = salt_size, n_ext = values_num_per_poly << rate_bits):
 *   [pad .. pad + (P+S)*n_ext)          column-major bit-reversed LDE         (plonky2_gpu.cu:461)
 *   [pad + (P+S)*n_ext .. + 4*num_digests + 4*2^cap_height)  digests || cap   (plonky2_gpu.cu:552)
 *   [0 .. (P+S)*n_ext)                  leaf-major LDE, written last after ctx->stream2 has been
 *                                       synchronised (plonky2_gpu.cu:586-591)
 * with pad = pad_extvalues_len. d_values_flatten holds the coefficients [P][n] and may alias
 * d_ext_values_flatten (plonky2/src/fri/oracle.rs:409-422 passes the same pointer).
 * d_shift_powers / root tables are accepted and ignored; the coset shift is 7. */
GlError merkle_tree_from_coeffs(uint64_t *d_values_flatten, uint64_t *d_ext_values_flatten, int poly_num,
                                int values_num_per_poly, int log_len, const uint64_t *d_root_table,
                                const uint64_t *d_root_table2, const uint64_t *d_shift_powers, int rate_bits,
                                int salt_size, int cap_height, int pad_extvalues_len, void *ctx);

/* lib.rs:83-98. The reference's definition is `assert(0)` (plonky2_gpu.cu:228); here it is the
 * working composition ifft + merkle_tree_from_coeffs. */
GlError merkle_tree_from_values(uint64_t *d_values_flatten, uint64_t *d_ext_values_flatten, int poly_num,
                                int values_num_per_poly, int log_len, const uint64_t *d_root_table,
                                const uint64_t *d_root_table2, const uint64_t *d_shift_powers, const uint64_t *n_inv,
                                int rate_bits, int salt_size, int cap_height, int pad_extvalues_len, void *ctx);

/* lib.rs:71-81 / plonky2_gpu.cu:138-189: Merkle tree over an LDE already resident at
 * d_ext_values_flatten + pad (column-major, NATURAL order as in the reference, which bit-reverses
 * it first): bit-reverses each column in place, then hashes and reduces as above. */
GlError build_merkle_tree(uint64_t *d_ext_values_flatten, int poly_num, int values_num_per_poly, int log_len,
                          int rate_bits, int salt_size, int cap_height, int pad_extvalues_len, void *ctx);

/* lib.rs:117-143 / plonky2_gpu.cu:609-783: the quotient polynomials of the ONE circuit the reference kernel is
 * hard-wired to — plonky2-ed25519: 234 wires / 80 routed, 8 constants, 2 challenges, quotient degree factor 8 =
 * 2^rate_bits, 25 gates in 6 selector groups, 231 gate constraints (plonky2_gpu.cu:666-675,
 * plonky2_gpu_impl.cuh:597-685). Same contract as the reference:
 *   - d_ext_values_flatten (wires), zs_partial_products_commitment_leaves->ptr and
 *     constants_sigmas_commitment_leaves->ptr are DEVICE buffers of LEAF-MAJOR rows [n_ext][leaf_len] with leaf t
 *     holding the point bitrev(t) (leaf_len 234 + salt_size, 20, 88; n_ext = values_num_per_poly << rate_bits) —
 *     region A of merkle_tree_from_values/coeffs; the DataSlice lengths are checked as the reference asserts them;
 *   - k_is (>= 80), alphas, betas, gammas (2 each) are DEVICE slices;
 *   - d_outs [2][n_ext] is scratch, d_quotient_polys [2][n_ext] receives the COEFFICIENTS of the two quotient
 *     polynomials (values on the coset -> ifft -> x shift^-i, plonky2_gpu.cu:737-765); the call synchronises ctx->stream.
 * Not read: d_root_table2, d_shift_inv_powers, points, z_h_on_coset_evals, z_h_on_coset_inverses (may be NULL) —
 * the kernels derive these from the library's own tables. Other shapes return GL_E_INVALID: every other circuit goes
 * through gl_compute_quotient_polys, of which this is one instance with the gate programs compiled in
 * (csrc/ed25519_gate_program.inc). The first call on a device builds its kernels with hiprtc: about a minute
 * of compilation when nothing is cached (six units; 15 s on the 8-core build container when forked workers compile them side by
 * side), about 2 s when ROCm's own compilation cache (~/.cache/comgr, on by default) has seen the
 * sources, immediate from
 * $PLONKY2_HIP_KERNEL_CACHE or, when that is unset, from the directory kernel_cache/ next to this library, where the
 * build (__graft_entry__.build()) puts the precompiled code objects. gl_reference_quotient_prepare() does it ahead of
 * the first proof.
 * The reference also compiles in the hash of one proof's public inputs (plonky2_gpu.cu:686-689); the same value is
 * the default here; gl_reference_set_public_inputs_hash_ctx() sets the value for the calls of ONE context (two circuits in one
 * process, each proving on its own context, do not race), gl_reference_set_public_inputs_hash() the process-wide value that
 * contexts without their own use (NULL restores the default in both).
 * Device memory: the kernels read column-major data 2.5x faster than leaf-major rows (a 1872-byte stride between
 * the lanes of a wave), so the call first transposes the three inputs into a staging buffer of
 * gl_reference_quotient_staging_bytes(log_len) = (234 + 88 + 20) * n_ext * 8 bytes per device (5.7 GB at log_len 18; salt
 * columns add salt_size * n_ext * 8). Either the CALLER provides it — gl_reference_quotient_set_staging(d_ptr, bytes) on the
 * current device, the reference's contract of a callee that allocates nothing; a buffer too small for a call is not used —
 * or the library allocates it on first use, keeps it for the next proof and frees it in gl_reference_quotient_release().
 * Without a staging buffer (allocation failed, too small, or PLONKY2_HIP_REFERENCE_IN_PLACE=1 in the environment) the
 * rows are read in place (same result, slower). Calls of compute_quotient_polys on one device take turns. */
GlError gl_reference_quotient_prepare(void *ctx);
GlError gl_reference_quotient_release(void);
uint64_t gl_reference_quotient_staging_bytes(int log_len);
GlError gl_reference_quotient_set_staging(void *d_staging /* NULL: library-owned again */, uint64_t bytes);
GlError gl_reference_set_public_inputs_hash(const uint64_t *h_hash /* 4, host */);
GlError gl_reference_set_public_inputs_hash_ctx(const uint64_t *h_hash /* 4, host */, void *ctx);
GlError compute_quotient_polys(const uint64_t *d_ext_values_flatten, int poly_num, int values_num_per_poly,
                               int log_len, const uint64_t *d_root_table2, const uint64_t *d_shift_inv_powers,
                               int rate_bits, int salt_size, const GlDataSlice *zs_partial_products_commitment_leaves,
                               const GlDataSlice *constants_sigmas_commitment_leaves, void *d_outs,
                               void *d_quotient_polys, const GlDataSlice *points, const GlDataSlice *z_h_on_coset_evals,
                               const GlDataSlice *z_h_on_coset_inverses, const GlDataSlice *k_is,
                               const GlDataSlice *alphas, const GlDataSlice *betas, const GlDataSlice *gammas, void *ctx);

/* The Rust wrapper falls back to this symbol when an Error carries no message
 * (cuda/src/lib.rs:42-45). Forwards to hipGetErrorString. */
const char *cudaGetErrorString(int code);

/* Test hook: element-wise field op on device arrays (op: 0 add, 1 sub, 2 mul, 3 neg, 4 x^7,
 * 5 a + b*b, 6 a * 2^(b mod 192), 7 a + canon(b), 8-16 internal variants, 17 a + (b mod 2^63) * 2^32, 18-27 the grouped
 * forms with deferred corrections); output canonical. d_b may be NULL for unary ops. Ops 100-109: the register-level radix
 * routines of the NTT passes on vectors of sixteen elements (n = 16 x vectors; d_b unused), see csrc/capi.hip. */
GlError gl_debug_field_op(int op, const uint64_t *d_a, const uint64_t *d_b, uint64_t *d_out, uint64_t n, void *ctx);
/* Measurement hook: a plain streaming copy kernel (16 B per lane, asynchronous on ctx->stream) — the bandwidth a
 * kernel can actually reach on this device, which bench.py reports next to the 8 TB/s specification. */
GlError gl_debug_copy(void *d_dst, const void *d_src, uint64_t bytes, void *ctx);

/* Library identification: "plonky2_hip <version> gfx950". */
const char *gl_version(void);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* PLONKY2_HIP_H */
