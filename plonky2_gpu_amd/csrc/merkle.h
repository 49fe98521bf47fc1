// merkle.h — internal interface of the Poseidon / Merkle / transpose kernels (see merkle.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace plonky2_hip {

// MerkleTree::new over leaves given column-major: cols[j*col_stride + i] = element j of leaf i.
// digests: 4*2*(n_leaves - 2^cap_height) u64 in the reference layout; cap: 4*2^cap_height u64.
// rows (optional): the leaf-major copy rows[i*leaf_len + j], written by the hashing lanes as they absorb (each lane has its
// leaf's elements in registers anyway; the 8-byte stores of a rate block are merged in L2 and cost the ALU-bound kernel nothing).
hipError_t merkle_tree_from_columns(const uint64_t *cols, uint32_t leaf_len, uint64_t n_leaves, uint64_t col_stride,
                                    uint32_t cap_height, uint64_t *digests, uint64_t *cap, hipStream_t stream, uint64_t *rows = nullptr);
// The leaf hashing of merkle_tree_from_columns cut at column boundaries: absorbs columns [c0, c1) of every leaf (c0 a
// multiple of 8; c1 - c0 a multiple of 8 unless c1 == leaf_len); the sponge's capacity travels between launches in the leaf's
// digest slot, the launch with c1 == leaf_len leaves the digest there. Then merkle_tree_layers builds the tree above.
// rows_from: the leaf-major copy is written for leaves i >= rows_from only (the caller transposes the others later: their slots
// in `rows` may hold the coefficients that the producer of later columns still reads, see commit_from_coeffs_impl).
hipError_t hash_leaves_chunk(const uint64_t *cols, uint32_t c0, uint32_t c1, uint32_t leaf_len, uint64_t n_leaves, uint64_t col_stride,
                             uint32_t cap_height, uint64_t *digests, uint64_t *cap, hipStream_t stream, uint64_t *rows = nullptr,
                             uint64_t rows_from = 0);
hipError_t merkle_tree_layers(uint64_t *digests, uint64_t *cap, uint64_t n_leaves, uint32_t cap_height, hipStream_t stream);
// Same for leaf-major rows[i*leaf_len + j].
hipError_t merkle_tree_from_rows(const uint64_t *rows, uint32_t leaf_len, uint64_t n_leaves, uint32_t cap_height,
                                 uint64_t *digests, uint64_t *cap, hipStream_t stream);
// states[count][12] permuted in place, canonical output.
hipError_t poseidon_permute_batch(uint64_t *states, uint64_t count, hipStream_t stream);
// state[12] <- overwrite-mode sponge over n_blocks full rate blocks of inputs (one lane, serial).
hipError_t sponge_absorb(uint64_t *d_state, const uint64_t *d_inputs, uint32_t n_blocks, hipStream_t stream);
// leaves + Merkle paths of `count` leaf indices: element j of leaf i at leaves[i*row_stride + j*elem_stride].
hipError_t merkle_open_batch(const uint64_t *leaves, uint64_t row_stride, uint64_t elem_stride, uint32_t leaf_len, uint64_t n_leaves,
                             uint32_t cap_height, const uint64_t *digests, const uint64_t *d_idx, uint32_t count, uint64_t *out_leaves,
                             uint64_t *out_sib, hipStream_t stream, uint64_t idx_mask = ~0ull, uint32_t idx_shift = 0);
// One step of the device-resident Challenger (merkle.hip challenger_step_kernel): d_challenger = 32 u64 (flags & 1: start from the
// empty transcript); observes the sources in order, then writes n_out challenges (flags & 2: the 4-word hash_no_pad instead) to d_out.
hipError_t challenger_step(uint64_t *d_challenger, const uint64_t *const *src_ptrs, const uint64_t *src_counts, const uint64_t *src_planar,
                           uint32_t n_src, uint32_t n_out, uint32_t flags, uint64_t *d_out, hipStream_t stream);
// cols[c*col_stride + r] -> rows[r*n_cols + c]
hipError_t transpose_to_leaf_major(const uint64_t *cols, uint64_t *rows, uint32_t n_cols, uint64_t n_rows,
                                   uint64_t col_stride, hipStream_t stream);
// rows[r*n_cols + c] -> cols[c*col_stride + r]
hipError_t transpose_to_column_major(const uint64_t *rows, uint64_t *cols, uint32_t n_cols, uint64_t n_rows, uint64_t col_stride,
                                     hipStream_t stream);

}  // namespace plonky2_hip
