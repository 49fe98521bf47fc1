// plonk.h — internal interface of the permutation-argument kernels (see plonk.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gate_jit.h"
#include "ntt.h"

namespace plonky2_hip {

uint32_t num_partial_products(uint32_t num_routed, uint32_t degree);

// out: [num_challenges * (1 + num_prods)][n] column-major in the reference's zs_partial_products order
// (all Z first, then the partial products challenge-major; plonky2/src/plonk/prover.rs:106-117).
hipError_t permutation_partial_products(const NttTables &tb, const uint64_t *wires, uint64_t wires_stride, const uint64_t *sigmas,
                                        uint64_t sigmas_stride, const uint64_t *k_is, const uint64_t *betas, const uint64_t *gammas,
                                        uint32_t num_challenges, uint32_t num_routed, uint32_t degree, uint32_t log_n, uint64_t *out,
                                        hipStream_t stream);

struct GateProgramArgs {
    const uint16_t *instrs;  // device
    const uint32_t *gates;   // device
    const uint64_t *imms;    // device (may be null when no LOAD_IMM is used)
    uint32_t num_gates, num_selectors;
    uint64_t public_inputs_hash[4];
};

struct QuotientArgs {
    const GateProgramArgs *gate_program = nullptr;  // alternative to gate_terms
    const GateKernel *gate_kernel = nullptr;        // alternative to both: run-time compiled gates (gate_jit.h)
    uint64_t *gate_partial_workspace = nullptr;     // device [num_challenges][lde_size], needed with gate_kernel
    const uint64_t *public_inputs_hash = nullptr;   // host, 4, needed with gate_kernel
    uint64_t column_stride = 0;                     // 0: leaf-major rows; else column-major with this column stride
    const uint64_t *wires_leaves, *cs_leaves, *zpp_leaves;  // LDE of the three commitments in either layout
    uint32_t wires_len, cs_len, zpp_len;                    // leaf lengths
    const uint64_t *k_is;                                   // device, num_routed
    const uint64_t *gate_terms;                             // device [lde_size][num_gate_constraints] or null
    const uint64_t *betas, *gammas, *alphas;                // host, num_challenges each
    uint32_t num_constants, num_routed, num_challenges, num_gate_constraints;
    uint32_t degree_bits, rate_bits, quotient_degree_factor;
    uint64_t shift;
};

// out: [num_challenges][n << log2_ceil(quotient_degree_factor)] quotient VALUES on the coset (natural order)
hipError_t quotient_values(const NttTables &tb, const QuotientArgs &a, uint64_t *out, hipStream_t stream);

// out[(q*n_polys + poly)*2 + {0,1}] = poly(points[q]) in F_{p^2} (points[q] = (points[2q], points[2q+1]), host array)
hipError_t eval_polys_ext2(const NttTables &tb, const uint64_t *coeffs, uint64_t n_polys, uint32_t log_n, uint64_t stride,
                           const uint64_t *points, uint32_t n_points, uint64_t *out, hipStream_t stream);

}  // namespace plonky2_hip
