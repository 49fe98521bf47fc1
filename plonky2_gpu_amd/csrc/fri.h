// fri.h — internal interface of the FRI primitives (see fri.hip). Extension vectors are planar:
// v[0..len) = first components, v[len..2len) = second components.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ntt.h"

namespace plonky2_hip {

hipError_t fri_reduce_polys_base(const NttTables &tb, const uint64_t *const *d_poly_ptrs, uint32_t m, uint64_t n, const uint64_t alpha[2],
                                 uint64_t *d_out, hipStream_t stream);
// d_comp (planar, n) is destroyed. d_final (planar, n): final[0] = 0, final[i+1] = (accumulate ? final[i+1]*scale : 0) + q_i
hipError_t fri_divide_by_linear_accumulate(const NttTables &tb, uint64_t *d_comp, uint64_t n, const uint64_t z[2], const uint64_t scale[2],
                                           int accumulate, uint64_t *d_final, hipStream_t stream);
// d_beta (optional): beta in device memory (two canonical words) instead of the host's `beta`
hipError_t fri_fold(const uint64_t *d_coeffs, uint64_t len, uint32_t arity_bits, const uint64_t beta[2], uint64_t *d_out,
                    hipStream_t stream, const uint64_t *d_beta = nullptr);
hipError_t fri_interleave(const uint64_t *d_planes, uint64_t len, uint64_t *d_rows, hipStream_t stream);
// synchronous (returns the smallest witness)
// d_challenger (optional): the duplex state comes from a device-resident Challenger (32 words) instead of `state` / `pos`;
// d_witness (optional): the device word that receives the witness (else a word of the workspace)
hipError_t fri_proof_of_work(const NttTables &tb, const uint64_t state[12], uint32_t pos, uint32_t min_leading_zeros, uint64_t *witness,
                             hipStream_t stream, const uint64_t *d_challenger = nullptr, uint64_t *d_witness = nullptr);

}  // namespace plonky2_hip
