// merkle.hip — Poseidon leaf hashing, Merkle-cap construction and the leaf-major transpose.
//
// Replaces MerkleTree::new (plonky2/src/hash/merkle_tree.rs:283-319; fill_digests_buf :210-244,
// fill_subtree :78-105) together with the transpose + reverse_index_bits that precede it in
// PolynomialBatch::from_coeffs (plonky2/src/fri/oracle.rs:942-952), and the reference's
// hash_leaves_kernel / reduce_digests_kernel / transpose_kernel
// (cuda/plonky2_gpu_impl.cuh:349-479).
//
// Layout contract kept bit-for-bit: `digests` is the reference's recursive
// "left subtree | left digest | right digest | right subtree" array (merkle_tree.rs:46-54). We do
// not recurse: the closed form used by MerkleTree::prove (merkle_tree.rs:424-435) says the pair q
// of layer L lives at hash index 2*((q << (L+1)) + 2^L - 1) + parity inside its cap subtree, so
// every layer is a flat data-parallel launch over ALL cap subtrees at once (the reference's GPU
// code uses one 256-thread block per cap entry, i.e. 16 blocks for the whole chip).
//
// Leaves are hashed straight from the NTT's column-major output: thread i walks the columns of
// row i, so a wavefront reads 64 consecutive u64 (512 B) of one column per load instruction —
// fully coalesced without materialising the leaf-major matrix first.
#include "merkle.h"
#include "knobs.h"

#include <mutex>

#include "poseidon.h"
#include "poseidon_coop.h"

namespace plonky2_hip {

namespace {

struct alignas(16) u64x2 {
    uint64_t x, y;
};

__device__ __forceinline__ void store_hash(uint64_t *dst, const uint64_t (&s)[12]) {
    u64x2 a = {gl::canon(s[0]), gl::canon(s[1])}, b = {gl::canon(s[2]), gl::canon(s[3])};
    reinterpret_cast<u64x2 *>(dst)[0] = a;
    reinterpret_cast<u64x2 *>(dst)[1] = b;
}

// Eight consecutive elements of a leaf-major row from the lane that has just absorbed them. A lane's 64 bytes lie alone (the next
// lane's row is leaf_len * 8 bytes away), so what counts is the number of write requests: 16-byte pieces, with a single element
// before and after them where the row position is only 8-byte aligned (rows of odd length alternate) — 4 or 5 requests
// instead of 8.
__device__ __forceinline__ void store_row_block(uint64_t *dst, const uint64_t (&s)[12]) {
    if ((reinterpret_cast<uintptr_t>(dst) & 8) == 0) {
#pragma unroll
        for (int q = 0; q < 4; q++) reinterpret_cast<u64x2 *>(dst)[q] = u64x2{s[2 * q], s[2 * q + 1]};
    } else {
        dst[0] = s[0];
#pragma unroll
        for (int q = 0; q < 3; q++) reinterpret_cast<u64x2 *>(dst + 1)[q] = u64x2{s[2 * q + 1], s[2 * q + 2]};
        dst[7] = s[7];
    }
}

// hash index (in units of 4 u64) of node `idx` of layer L inside a cap subtree
__device__ __forceinline__ uint64_t digest_slot(uint64_t idx, uint32_t L) {
    uint64_t q = idx >> 1, parity = idx & 1;
    return 2 * ((q << (L + 1)) + (1ull << L) - 1) + parity;
}

// H::hash_or_noop on every row of a column-major matrix (plonky2/src/plonk/config.rs:56-67,
// hash/hashing.rs:81-108). cols[j*col_stride + i] = element j of leaf i.
__global__ __launch_bounds__(256) void hash_leaves_kernel(const uint64_t *__restrict__ cols, uint32_t leaf_len,
                                                          uint64_t n_leaves, uint64_t col_stride,
                                                          uint64_t *__restrict__ digests, uint64_t *__restrict__ cap,
                                                          uint32_t log_sub_leaves, uint64_t *__restrict__ rows) {
    const poseidon::MdsOperands ops = poseidon::mds_operands();
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    // every lane stays in step to the end (poseidon.h: the matrix instructions want whole waves): lanes past the end hash the
    // last leaf once more and store nothing
    const bool live = i < n_leaves;
    if (!live) i = n_leaves - 1;
    uint64_t s[12];
#pragma unroll
    for (int k = 0; k < 12; k++) s[k] = 0;
    uint64_t *row = rows && live ? rows + i * leaf_len : nullptr;  // the leaf-major copy, written by the lane that holds the leaf anyway
    if (leaf_len <= 4) {
        // not hashed: copied, zero padded, canonicalised by store_hash (config.rs:57-63)
#pragma unroll
        for (int k = 0; k < 4; k++)
            if ((uint32_t)k < leaf_len) {
                s[k] = cols[(uint64_t)k * col_stride + i];
                if (row) row[k] = s[k];
            }
    } else {
        uint32_t j = 0;
        for (; j + 8 <= leaf_len; j += 8) {
#pragma unroll
            for (int k = 0; k < 8; k++) s[k] = cols[(uint64_t)(j + k) * col_stride + i];
            if (row) store_row_block(row + j, s);
            poseidon::permute(s, ops);
        }
        if (j < leaf_len) {
            // short last chunk overwrites only its own lanes (hashing.rs:89-92)
#pragma unroll
            for (int k = 0; k < 8; k++)
                if (j + k < leaf_len) {
                    s[k] = cols[(uint64_t)(j + k) * col_stride + i];
                    if (row) row[j + k] = s[k];
                }
            poseidon::permute(s, ops);
        }
    }
    if (!live) return;
    if (log_sub_leaves == 0) {
        store_hash(cap + 4 * i, s);
    } else {
        uint64_t sub = i >> log_sub_leaves, idx = i & ((1ull << log_sub_leaves) - 1);
        uint64_t sub_digests = 2 * ((1ull << log_sub_leaves) - 1);
        store_hash(digests + 4 * (sub * sub_digests + digest_slot(idx, 0)), s);
    }
}

// The same sponge cut at column boundaries, so that hashing can start before the last columns exist (a commit's LDE
// produces them chunk by chunk): this launch absorbs columns [c0, c1) of every leaf, c0 a multiple of 8. In overwrite mode
// a full block replaces all eight rate words (hashing.rs:89-98), so what survives from one permutation to the next full
// block is the CAPACITY, four words — exactly the size of the leaf's digest slot, which carries it between launches
// (raw u64 representatives; the last launch overwrites it with the canonical digest). leaf_len > 4 (shorter leaves are
// not hashed at all, config.rs:57-63).
__global__ __launch_bounds__(256) void hash_leaves_chunk_kernel(const uint64_t *__restrict__ cols, uint32_t c0, uint32_t c1,
                                                                uint32_t leaf_len, uint64_t n_leaves, uint64_t col_stride,
                                                                uint64_t *__restrict__ digests, uint64_t *__restrict__ cap,
                                                                uint32_t log_sub_leaves, uint64_t *__restrict__ rows, uint64_t rows_from) {
    const poseidon::MdsOperands ops = poseidon::mds_operands();
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < n_leaves;  // see hash_leaves_kernel
    if (!live) i = n_leaves - 1;
    // rows below rows_from are not written here: their slots may still hold data that later chunks' producers read
    uint64_t *row = rows && live && i >= rows_from ? rows + i * leaf_len : nullptr;
    uint64_t *slot;
    if (log_sub_leaves == 0) {
        slot = cap + 4 * i;
    } else {
        uint64_t sub = i >> log_sub_leaves, idx = i & ((1ull << log_sub_leaves) - 1);
        uint64_t sub_digests = 2 * ((1ull << log_sub_leaves) - 1);
        slot = digests + 4 * (sub * sub_digests + digest_slot(idx, 0));
    }
    uint64_t s[12];
#pragma unroll
    for (int k = 0; k < 12; k++) s[k] = 0;
    if (c0 != 0) {
        const u64x2 a = reinterpret_cast<const u64x2 *>(slot)[0], b = reinterpret_cast<const u64x2 *>(slot)[1];
        s[8] = a.x, s[9] = a.y, s[10] = b.x, s[11] = b.y;
    }
    uint32_t j = c0;
    for (; j + 8 <= c1; j += 8) {
#pragma unroll
        for (int k = 0; k < 8; k++) s[k] = cols[(uint64_t)(j + k) * col_stride + i];
        if (row) store_row_block(row + j, s);
        poseidon::permute(s, ops);
    }
    if (c1 == leaf_len) {
        if (j < leaf_len) {
            // short last block overwrites only its own lanes (hashing.rs:89-92): the other rate words are the previous
            // permutation's, which this launch has just computed (c1 - c0 >= 8) or, for a chunk shorter than a block, cannot
            // have — the host never makes such a chunk (see merkle_tree_from_columns_chunked)
#pragma unroll
            for (int k = 0; k < 8; k++)
                if (j + k < leaf_len) {
                    s[k] = cols[(uint64_t)(j + k) * col_stride + i];
                    if (row) row[j + k] = s[k];
                }
            poseidon::permute(s, ops);
        }
        if (live) store_hash(slot, s);
    } else if (live) {
        reinterpret_cast<u64x2 *>(slot)[0] = u64x2{s[8], s[9]};
        reinterpret_cast<u64x2 *>(slot)[1] = u64x2{s[10], s[11]};
    }
}

// Same, for leaf-major input rows[i*leaf_len + j] (used by gl_merkle_tree_from_leaves).
// poseidon::permute with the s-boxes' rare corrections executed always instead of behind a branch (gl::pow7_nb, poseidon_coop.h): for
// the layers that leave a wave alone on its SIMD, where the permutation's latency is the layer's time.
__device__ __forceinline__ void permute_latency(uint64_t (&s)[12], const poseidon::MdsOperands &ops) {
    using poseidon::HALF_FULL;
    using poseidon::N_PARTIAL;
    constexpr int W = poseidon::W;
    poseidon::require_full_wave();
#pragma unroll
    for (int i = 0; i < W; i++) s[i] = gl::add_canonical(s[i], POSEIDON_ALL_ROUND_CONSTANTS[i]);
#pragma unroll 1
    for (int r = 0; r < 2 * HALF_FULL + N_PARTIAL; r++) {
        if (r < HALF_FULL || r >= HALF_FULL + N_PARTIAL) {
#pragma unroll
            for (int i = 0; i < W; i++) s[i] = gl::pow7_nb(s[i]);
        } else {
            s[0] = gl::pow7_nb(s[0]);
        }
        poseidon::mds_layer(s, ops, POSEIDON_MDS_XY + 2 * W * r);
    }
}

template <bool LATENCY>  // at most one wave per SIMD: the permutation with the branch-free s-box (permute_latency)
__global__ __launch_bounds__(256) void hash_rows_kernel(const uint64_t *__restrict__ rows, uint32_t leaf_len,
                                                        uint64_t n_leaves, uint64_t *__restrict__ digests,
                                                        uint64_t *__restrict__ cap, uint32_t log_sub_leaves) {
    const poseidon::MdsOperands ops = poseidon::mds_operands();
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < n_leaves;  // see hash_leaves_kernel
    if (!live) i = n_leaves - 1;
    const uint64_t *row = rows + i * leaf_len;
    uint64_t s[12];
#pragma unroll
    for (int k = 0; k < 12; k++) s[k] = 0;
    if (leaf_len <= 4) {
#pragma unroll
        for (int k = 0; k < 4; k++)
            if ((uint32_t)k < leaf_len) s[k] = row[k];
    } else {
        uint32_t j = 0;
        for (; j + 8 <= leaf_len; j += 8) {
#pragma unroll
            for (int k = 0; k < 8; k++) s[k] = row[j + k];
            LATENCY ? permute_latency(s, ops) : poseidon::permute(s, ops);
        }
        if (j < leaf_len) {
#pragma unroll
            for (int k = 0; k < 8; k++)
                if (j + k < leaf_len) s[k] = row[j + k];
            LATENCY ? permute_latency(s, ops) : poseidon::permute(s, ops);
        }
    }
    if (!live) return;
    if (log_sub_leaves == 0) {
        store_hash(cap + 4 * i, s);
    } else {
        uint64_t sub = i >> log_sub_leaves, idx = i & ((1ull << log_sub_leaves) - 1);
        uint64_t sub_digests = 2 * ((1ull << log_sub_leaves) - 1);
        store_hash(digests + 4 * (sub * sub_digests + digest_slot(idx, 0)), s);
    }
}

// One tree layer for all cap subtrees: parent = two_to_one(left, right) (hashing.rs:65-72).
// Thread g handles pair q = g mod pairs_per_sub of subtree g / pairs_per_sub at layer L. LATENCY: the layer has at most one wave per SIMD.
template <bool LATENCY>
__global__ __launch_bounds__(256) void tree_layer_kernel(uint64_t *__restrict__ digests, uint64_t *__restrict__ cap,
                                                         uint32_t L, uint32_t log_sub_leaves, uint64_t total_pairs) {
    const poseidon::MdsOperands ops = poseidon::mds_operands();
    uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = g < total_pairs;  // see hash_leaves_kernel
    if (!live) g = total_pairs - 1;
    uint32_t log_pairs = log_sub_leaves - L - 1;
    uint64_t sub = g >> log_pairs, q = g & ((1ull << log_pairs) - 1);
    uint64_t sub_digests = 2 * ((1ull << log_sub_leaves) - 1);
    uint64_t *tree = digests + 4 * sub * sub_digests;
    const u64x2 *pair = reinterpret_cast<const u64x2 *>(tree + 4 * digest_slot(2 * q, L));
    uint64_t s[12];
    u64x2 a = pair[0], b = pair[1], c = pair[2], d = pair[3];
    s[0] = a.x; s[1] = a.y; s[2] = b.x; s[3] = b.y;
    s[4] = c.x; s[5] = c.y; s[6] = d.x; s[7] = d.y;
    s[8] = s[9] = s[10] = s[11] = 0;
    if (LATENCY)
        permute_latency(s, ops);
    else
        poseidon::permute(s, ops);
    if (!live) return;
    if (log_pairs == 0)
        store_hash(cap + 4 * sub, s);
    else
        store_hash(tree + 4 * digest_slot(q, L + 1), s);
}

__global__ __launch_bounds__(256) void permute_batch_kernel(uint64_t *states, uint64_t count) {
    const poseidon::MdsOperands ops = poseidon::mds_operands();
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < count;  // see hash_leaves_kernel
    if (!live) i = count - 1;
    uint64_t s[12];
#pragma unroll
    for (int k = 0; k < 12; k++) s[k] = states[i * 12 + k];
    poseidon::permute(s, ops);
    if (!live) return;
#pragma unroll
    for (int k = 0; k < 12; k++) states[i * 12 + k] = gl::canon(s[k]);
}

#ifdef PLONKY2_DEBUG_KNOBS
// The diagnostic build answers gl_poseidon_permute_batch from the vector-ALU permutation under PLONKY2_POSEIDON=vector: the second,
// independent implementation (poseidon_vector.h) that tests/test_gpu_merkle.py holds against the matrix-core one.
__global__ __launch_bounds__(256) void permute_batch_vector_kernel(uint64_t *states, uint64_t count) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    uint64_t s[12];
#pragma unroll
    for (int k = 0; k < 12; k++) s[k] = states[i * 12 + k];
    poseidon_vector::permute(s);
#pragma unroll
    for (int k = 0; k < 12; k++) states[i * 12 + k] = gl::canon(s[k]);
}
#endif

// The transcript's sponge (iop/challenger.rs:131-149 run over several full rate blocks): serial by
// definition, so ONE wavefront computes each permutation cooperatively (poseidon_coop.h); state[0..8)
// is overwritten by each block, then permuted.
__global__ __launch_bounds__(64) void sponge_absorb_kernel(uint64_t *state, const uint64_t *inputs, uint32_t n_blocks,
                                                           poseidon_coop::Tables tb) {
    __shared__ uint64_t lds[12];
    const int lane = threadIdx.x;
    uint64_t x = lane < 12 ? state[lane] : 0;
    for (uint32_t b = 0; b < n_blocks; b++) {
        if (lane < 8) x = inputs[8 * b + lane];
        x = poseidon_coop::permute(x, tb, lds);
    }
    if (lane < 12) state[lane] = gl::canon(x);
}

// The Challenger with its state in DEVICE memory (iop/challenger.rs:19-149): T[0..12) sponge state, T[12..20) input buffer,
// T[28] its length, T[29] the number of unread outputs (the output buffer itself is state[0..8) while that is non-zero: the state
// does not change between a duplexing and the next one). One launch = observe_elements over up to eight sources, in order, read
// where the producing kernels left them (a cap, the openings, planar extension coefficients), then get_n_challenges — no host
// round trip inside a transcript step; the host fetches only the challenges it needs itself.
struct ChallengerSrc {
    const uint64_t *p;
    uint64_t count;       // elements of this source
    uint64_t planar_len;  // != 0: element i is p[(i & 1) * planar_len + (i >> 1)] (an extension vector kept as two planes)
};
struct ChallengerArgs {
    ChallengerSrc src[8];
    uint32_t n_src, n_out, flags;
};
constexpr uint32_t CH_RESET = 1, CH_HASH = 2;

__global__ __launch_bounds__(64) void challenger_step_kernel(uint64_t *__restrict__ T, ChallengerArgs a, uint64_t *__restrict__ out,
                                                             poseidon_coop::Tables tb) {
    __shared__ uint64_t lds[12];
    const int lane = threadIdx.x;
    const bool reset = a.flags & CH_RESET;
    uint64_t x = (lane < 12 && !reset) ? T[lane] : 0;           // lane k < 12: state word k
    uint64_t inb = (lane < 8 && !reset) ? T[12 + lane] : 0;     // lane k < 8: input buffer slot k
    uint32_t in_len = reset ? 0u : (uint32_t)T[28], out_len = reset ? 0u : (uint32_t)T[29];
    uint64_t total = 0;
    for (uint32_t i = 0; i < a.n_src; i++) total += a.src[i].count;
    auto fetch = [&](uint64_t g) -> uint64_t {  // element g of the concatenated sources, canonical (observe_element takes field elements)
        for (uint32_t i = 0; i < a.n_src; i++) {
            if (g < a.src[i].count) {
                const ChallengerSrc &sr = a.src[i];
                return gl::canon(sr.planar_len ? sr.p[(g & 1) * sr.planar_len + (g >> 1)] : sr.p[g]);
            }
            g -= a.src[i].count;
        }
        return 0;
    };
    uint64_t pos = 0;
    // observe_element (challenger.rs:43-53) for every element: the buffer fills, duplexing() overwrites the state's first words with it
    // and permutes (:131-149); every observation clears the output buffer, the eighth of a block refills it
    while (in_len + (total - pos) >= 8) {
        if (lane < 8) x = (uint32_t)lane < in_len ? inb : fetch(pos + lane - in_len);
        x = poseidon_coop::permute(x, tb, lds);
        pos += 8 - in_len;
        in_len = 0;
        out_len = 8;
    }
    if (pos < total) {
        const uint32_t r = (uint32_t)(total - pos);
        if ((uint32_t)lane >= in_len && (uint32_t)lane < in_len + r) inb = fetch(pos + lane - in_len);
        in_len += r;
        out_len = 0;
    }
    auto duplexing = [&]() {
        if ((uint32_t)lane < in_len) x = inb;
        x = poseidon_coop::permute(x, tb, lds);
        in_len = 0;
        out_len = 8;
    };
    if (a.flags & CH_HASH) {
        // hash_n_to_hash_no_pad (hash/hashing.rs:81-108): a short last chunk overwrites its own words and is permuted; out = state[0..4)
        if (in_len) duplexing();
        if (lane < 4) out[lane] = gl::canon(x);
    } else {
        for (uint32_t c = 0; c < a.n_out; c++) {  // get_challenge (:87-97): the output buffer is popped from the back
            if (in_len || !out_len) duplexing();
            const uint64_t v = poseidon_coop::lane_value(x, (int)out_len - 1);
            if (lane == 0) out[c] = gl::canon(v);
            out_len--;
        }
    }
    if (lane < 12) T[lane] = gl::canon(x);
    if (lane < 8) T[12 + lane] = inb;
    if (lane == 0) T[28] = in_len, T[29] = out_len;
}

// A tree layer with few nodes: one wavefront per node (latency ~6x shorter than a lane per node).
__global__ __launch_bounds__(64) void tree_layer_coop_kernel(uint64_t *__restrict__ digests, uint64_t *__restrict__ cap, uint32_t L,
                                                             uint32_t log_sub_leaves, poseidon_coop::Tables tb) {
    __shared__ uint64_t lds[12];
    const int lane = threadIdx.x;
    const uint64_t g = blockIdx.x;
    const uint32_t log_pairs = log_sub_leaves - L - 1;
    const uint64_t sub = g >> log_pairs, q = g & ((1ull << log_pairs) - 1);
    const uint64_t sub_digests = 2 * ((1ull << log_sub_leaves) - 1);
    uint64_t *tree = digests + 4 * sub * sub_digests;
    const uint64_t *pair = tree + 4 * digest_slot(2 * q, L);  // left digest | right digest, contiguous
    uint64_t x = lane < 8 ? pair[lane] : 0;
    x = poseidon_coop::permute(x, tb, lds);
    uint64_t *dst = log_pairs == 0 ? cap + 4 * sub : tree + 4 * digest_slot(q, L + 1);
    if (lane < 4) dst[lane] = gl::canon(x);
}

// MerkleTree::prove (merkle_tree.rs:392-440) + the leaf itself for `count` leaf indices at once:
// block q copies leaf idx[q] (leaf-major rows or column-major columns) and its sibling digests.
__global__ __launch_bounds__(64) void merkle_open_kernel(const uint64_t *__restrict__ leaves, uint64_t row_stride, uint64_t elem_stride,
                                                         uint32_t leaf_len, const uint64_t *__restrict__ digests, uint32_t num_layers,
                                                         uint64_t subtree_digests, const uint64_t *__restrict__ idx, uint64_t idx_mask,
                                                         uint32_t idx_shift, uint64_t *__restrict__ out_leaves, uint64_t *__restrict__ out_sib) {
    // the leaf of query q: (idx[q] & idx_mask) >> idx_shift — a query index is a challenge reduced to the LDE's size, and the index
    // into a FRI layer's tree is that shifted by the arities so far (fri/prover.rs:186-236); callers with plain indices pass ~0, 0
    const uint64_t q = blockIdx.x, leaf = (idx[q] & idx_mask) >> idx_shift;
    for (uint32_t j = threadIdx.x; j < leaf_len; j += blockDim.x) out_leaves[q * leaf_len + j] = leaves[leaf * row_stride + j * elem_stride];
    const uint64_t base = subtree_digests * (leaf >> num_layers);
    for (uint32_t t = threadIdx.x; t < 4 * num_layers; t += blockDim.x) {
        uint32_t l = t >> 2, k = t & 3;
        uint64_t node = (leaf & ((1ull << num_layers) - 1)) >> l;  // this path's node in layer l
        uint64_t slot = base + digest_slot(node ^ 1, l);
        out_sib[(q * num_layers + l) * 4 + k] = digests[4 * slot + k];
    }
}

// [n_cols][col_stride] column-major -> [n_rows][n_cols] leaf-major through a 64x64 LDS tile
// (+1 pad): both the global read (along rows of a column) and the global write (along columns of
// a row) are 512 B contiguous per wavefront.
constexpr int TP = 64;
__global__ __launch_bounds__(256) void transpose_kernel(const uint64_t *__restrict__ cols, uint64_t *__restrict__ rows,
                                                        uint32_t n_cols, uint64_t n_rows, uint64_t col_stride) {
    __shared__ uint64_t tile[TP][TP + 1];
    uint64_t r0 = (uint64_t)blockIdx.x * TP;
    uint32_t c0 = blockIdx.y * TP;
    uint32_t tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 64 x 4
#pragma unroll
    for (int k = 0; k < TP; k += 4) {
        uint32_t c = c0 + ty + k;
        uint64_t r = r0 + tx;
        if (c < n_cols && r < n_rows) tile[ty + k][tx] = cols[(uint64_t)c * col_stride + r];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < TP; k += 4) {
        uint64_t r = r0 + ty + k;
        uint32_t c = c0 + tx;
        if (c < n_cols && r < n_rows) rows[r * n_cols + c] = tile[tx][ty + k];
    }
}

// The way back: [n_rows][n_cols] leaf-major -> [n_cols][col_stride] column-major, same tile, same access widths.
__global__ __launch_bounds__(256) void transpose_back_kernel(const uint64_t *__restrict__ rows, uint64_t *__restrict__ cols,
                                                             uint32_t n_cols, uint64_t n_rows, uint64_t col_stride) {
    __shared__ uint64_t tile[TP][TP + 1];
    uint64_t r0 = (uint64_t)blockIdx.x * TP;
    uint32_t c0 = blockIdx.y * TP;
    uint32_t tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < TP; k += 4) {
        uint64_t r = r0 + ty + k;
        uint32_t c = c0 + tx;
        if (c < n_cols && r < n_rows) tile[ty + k][tx] = rows[r * n_cols + c];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < TP; k += 4) {
        uint32_t c = c0 + ty + k;
        uint64_t r = r0 + tx;
        if (c < n_cols && r < n_rows) cols[(uint64_t)c * col_stride + r] = tile[tx][ty + k];
    }
}

// ---------------------------------------------------------------------------------------------
// The same two transposes on strips: a workgroup moves 64 rows x a chunk of up to 96 columns (all of them when the matrix
// has no more). In the leaf-major matrix such a strip is contiguous when it spans all columns, and runs of chunk * 8 bytes per
// row otherwise, so that side is read or written as a FLAT range, 16 bytes per lane when whole; the column-major side is 512
// contiguous bytes of one column per wavefront instruction. The 64 x 64 tiles above leave 7 lanes in 64 busy on the third
// column tile of a 135-column matrix and never use 16-byte accesses (3.2 TB/s); the strips are kept for every shape.
// ---------------------------------------------------------------------------------------------
constexpr int STRIP_MAX_COLS = 96;    // 64-row strips: 64 x (pitch <= 97) x 8 bytes = 49 KB of LDS at most, three workgroups per CU
constexpr int STRIP_MAX_COLS32 = 192;  // 32-row strips for wider matrices: whole rows up to 192 columns in the same 49 KB

struct StripGeom {
    uint32_t n_cols, chunk, n_chunks, pitch;  // chunk columns per strip (the last strip may be shorter), LDS row pitch (odd)
    uint32_t magic, magic_last;               // floor(2^32 / count) + 1 for a full and for the last strip: f / count = (f * magic) >> 32
};

// f / count for f < 64 * 193. count == 1 has no 32-bit reciprocal (2^32 / 1 does not fit; the truncated magic would be 1 and
// every quotient 0): the quotient is f itself.
__device__ __forceinline__ uint32_t strip_div(uint32_t f, uint32_t count, uint32_t magic) {
    return count == 1 ? f : (uint32_t)(((uint64_t)f * magic) >> 32);
}

__device__ __forceinline__ void strip_range(const StripGeom &g, uint32_t &c_begin, uint32_t &c_count, uint32_t &magic) {
    c_begin = blockIdx.y * g.chunk;
    const bool last = g.n_cols - c_begin < g.chunk;
    c_count = last ? g.n_cols - c_begin : g.chunk;
    magic = last ? g.magic_last : g.magic;
}

// ROWS = 64: a wavefront instruction moves 64 rows of one column; ROWS = 32: 32 rows of two adjacent columns
template <int ROWS>
__global__ __launch_bounds__(256) void transpose_strip_kernel(const uint64_t *__restrict__ cols, uint64_t *__restrict__ rows, uint64_t n_rows,
                                                              uint64_t col_stride, const StripGeom g) {
    extern __shared__ __attribute__((aligned(16))) uint64_t strip[];
    constexpr uint32_t CPW = 64 / ROWS;  // columns per wavefront instruction
    const uint64_t r0 = (uint64_t)blockIdx.x * ROWS;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lr = lane & (ROWS - 1), lc = lane / ROWS;
    uint32_t c_begin, c_count, magic;
    strip_range(g, c_begin, c_count, magic);
    const bool row_ok = r0 + lr < n_rows;
    for (uint32_t c = wave * CPW + lc; c < c_count; c += 16 * CPW) {
        uint64_t v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t cc = c + 4 * CPW * k;
            v[k] = (cc < c_count && row_ok) ? cols[(uint64_t)(c_begin + cc) * col_stride + r0 + lr] : 0;
        }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t cc = c + 4 * CPW * k;
            if (cc < c_count) strip[lr * g.pitch + cc] = v[k];
        }
    }
    __syncthreads();
    // LDS -> rows, flat over the strip. A strip that spans all columns and whole rows is one contiguous, 16-byte aligned range
    // (ROWS is even and r0 a multiple of it), whatever the parity of the column count: the two elements of a piece may sit
    // in different rows.
    const uint32_t total = ROWS * c_count;
    if (c_count == g.n_cols && r0 + ROWS <= n_rows && (reinterpret_cast<uintptr_t>(rows) & 15) == 0) {  // an 8-byte aligned matrix takes the scalar path
        u64x2 *out = reinterpret_cast<u64x2 *>(rows + r0 * g.n_cols);
        for (uint32_t q = threadIdx.x; 2 * q < total; q += 256) {
            const uint32_t f = 2 * q, ra = strip_div(f, c_count, magic), ca = f - ra * c_count;
            const uint32_t rb = ca + 1 == c_count ? ra + 1 : ra, cb = ca + 1 == c_count ? 0 : ca + 1;
            u64x2 t;
            t.x = strip[ra * g.pitch + ca];
            t.y = strip[rb * g.pitch + cb];
            out[q] = t;
        }
    } else {
        for (uint32_t f = threadIdx.x; f < total; f += 256) {
            const uint32_t r = strip_div(f, c_count, magic), c = f - r * c_count;
            if (r0 + r < n_rows) rows[(r0 + r) * g.n_cols + c_begin + c] = strip[r * g.pitch + c];
        }
    }
}

template <int ROWS>
__global__ __launch_bounds__(256) void transpose_strip_back_kernel(const uint64_t *__restrict__ rows, uint64_t *__restrict__ cols, uint64_t n_rows,
                                                                   uint64_t col_stride, const StripGeom g) {
    extern __shared__ __attribute__((aligned(16))) uint64_t strip[];
    constexpr uint32_t CPW = 64 / ROWS;
    const uint64_t r0 = (uint64_t)blockIdx.x * ROWS;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lr = lane & (ROWS - 1), lc = lane / ROWS;
    uint32_t c_begin, c_count, magic;
    strip_range(g, c_begin, c_count, magic);
    const uint32_t total = ROWS * c_count;
    if (c_count == g.n_cols && r0 + ROWS <= n_rows && (reinterpret_cast<uintptr_t>(rows) & 15) == 0) {  // an 8-byte aligned matrix takes the scalar path
        const u64x2 *in = reinterpret_cast<const u64x2 *>(rows + r0 * g.n_cols);
        for (uint32_t q = threadIdx.x; 2 * q < total; q += 256) {
            const uint32_t f = 2 * q, ra = strip_div(f, c_count, magic), ca = f - ra * c_count;
            const uint32_t rb = ca + 1 == c_count ? ra + 1 : ra, cb = ca + 1 == c_count ? 0 : ca + 1;
            const u64x2 t = in[q];
            strip[ra * g.pitch + ca] = t.x;
            strip[rb * g.pitch + cb] = t.y;
        }
    } else {
        for (uint32_t f = threadIdx.x; f < total; f += 256) {
            const uint32_t r = strip_div(f, c_count, magic), c = f - r * c_count;
            if (r0 + r < n_rows) strip[r * g.pitch + c] = rows[(r0 + r) * g.n_cols + c_begin + c];
        }
    }
    __syncthreads();
    if (r0 + lr < n_rows)
        for (uint32_t c = wave * CPW + lc; c < c_count; c += 4 * CPW) cols[(uint64_t)(c_begin + c) * col_stride + r0 + lr] = strip[lr * g.pitch + c];
}

// 64-row strips up to 96 columns, 32-row strips beyond
static bool strip_rows32(uint32_t n_cols) { return n_cols > (uint32_t)STRIP_MAX_COLS; }

static StripGeom strip_geom(uint32_t n_cols) {
    const uint32_t max_cols = strip_rows32(n_cols) ? STRIP_MAX_COLS32 : STRIP_MAX_COLS;
    StripGeom g;
    g.n_cols = n_cols;
    g.n_chunks = (n_cols + max_cols - 1) / max_cols;
    g.chunk = (n_cols + g.n_chunks - 1) / g.n_chunks;
    g.n_chunks = (n_cols + g.chunk - 1) / g.chunk;
    g.pitch = g.chunk | 1;
    const uint32_t last = n_cols - (g.n_chunks - 1) * g.chunk;
    g.magic = (uint32_t)(0x100000000ull / g.chunk) + 1;  // exact for f * count < 2^32; f < 64 * 193
    g.magic_last = (uint32_t)(0x100000000ull / last) + 1;
    return g;
}

unsigned grid_for(uint64_t n, unsigned block) { return (unsigned)((n + block - 1) / block); }

// The leaves of a small tree (the later commit-phase trees of FRI: 2^9, 2^5 leaves of 32 elements): one wavefront per leaf, the sponge
// of hash_rows_kernel with the cooperative permutation — four permutations of 8 us instead of four of 40.
__global__ __launch_bounds__(64) void hash_rows_coop_kernel(const uint64_t *__restrict__ rows, uint32_t leaf_len, uint64_t *__restrict__ digests,
                                                            uint64_t *__restrict__ cap, uint32_t log_sub_leaves, poseidon_coop::Tables tb) {
    __shared__ uint64_t lds[12];
    const int lane = threadIdx.x;
    const uint64_t i = blockIdx.x;
    const uint64_t *row = rows + i * leaf_len;
    uint64_t x = 0;  // lane k < 12 holds state word k
    for (uint32_t j = 0; j < leaf_len; j += 8) {  // leaf_len > 4 (hash_or_noop's other case stays with hash_rows_kernel)
        if (lane < 8 && j + lane < leaf_len) x = row[j + lane];  // overwrite mode: a short last chunk leaves the old words
        x = poseidon_coop::permute(x, tb, lds);
    }
    uint64_t *dst;
    if (log_sub_leaves == 0) {
        dst = cap + 4 * i;
    } else {
        const uint64_t sub = i >> log_sub_leaves, idx = i & ((1ull << log_sub_leaves) - 1);
        const uint64_t sub_digests = 2 * ((1ull << log_sub_leaves) - 1);
        dst = digests + 4 * (sub * sub_digests + digest_slot(idx, 0));
    }
    if (lane < 4) dst[lane] = gl::canon(x);
}

// per-device tables of the cooperative permutation, built on first use
constexpr int MAX_DEVICES = 64;
std::mutex g_coop_mutex;
uint64_t *g_coop_tables[MAX_DEVICES] = {};

hipError_t coop_tables(poseidon_coop::Tables *tb, hipStream_t stream) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= MAX_DEVICES) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(g_coop_mutex);
    if (!g_coop_tables[dev]) {
        uint64_t *p = nullptr;
        const size_t elems = (size_t)(poseidon_coop::T0_ROWS + poseidon_coop::T_ROWS) * poseidon_coop::LANES;
        e = hipMalloc(&p, elems * sizeof(uint64_t));
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(poseidon_coop::build_tables_kernel, dim3(1), dim3(64), 0, stream, p,
                           p + poseidon_coop::T0_ROWS * poseidon_coop::LANES);
        e = hipStreamSynchronize(stream);
        if (e != hipSuccess) {
            (void)hipFree(p);
            return e;
        }
        g_coop_tables[dev] = p;
    }
    tb->t0 = g_coop_tables[dev];
    tb->t = g_coop_tables[dev] + poseidon_coop::T0_ROWS * poseidon_coop::LANES;
    return hipSuccess;
}

// layers with fewer nodes than this cannot fill the chip with one lane per node: they are latency
// bound, and a wavefront per node shortens the latency
constexpr uint64_t COOP_LAYER_MAX_NODES = 4096;
// ... and up to one wave per SIMD (1024 SIMDs x 64 lanes) the lane-per-node kernel runs with the branch-free s-box
constexpr uint64_t LATENCY_LAYER_MAX_NODES = 65536;

hipError_t tree_layers(uint64_t *digests, uint64_t *cap, uint64_t n_leaves, uint32_t log_sub_leaves, hipStream_t stream) {
    uint64_t n_sub = n_leaves >> log_sub_leaves;
    poseidon_coop::Tables tb = {};
    for (uint32_t L = 0; L < log_sub_leaves; L++) {
        uint64_t total = n_sub << (log_sub_leaves - L - 1);
        if (total <= COOP_LAYER_MAX_NODES) {
            if (!tb.t) {
                hipError_t e = coop_tables(&tb, stream);
                if (e != hipSuccess) return e;
            }
            hipLaunchKernelGGL(tree_layer_coop_kernel, dim3((unsigned)total), dim3(64), 0, stream, digests, cap, L, log_sub_leaves, tb);
        } else {
            if (total <= LATENCY_LAYER_MAX_NODES)
                hipLaunchKernelGGL(tree_layer_kernel<true>, dim3(grid_for(total, 256)), dim3(256), 0, stream, digests, cap, L, log_sub_leaves, total);
            else
                hipLaunchKernelGGL(tree_layer_kernel<false>, dim3(grid_for(total, 256)), dim3(256), 0, stream, digests, cap, L, log_sub_leaves, total);
        }
    }
    return hipGetLastError();
}

}  // namespace

static int log2_exact(uint64_t n) {
    int l = 0;
    while ((1ull << l) < n) l++;
    return (1ull << l) == n ? l : -1;
}

hipError_t merkle_tree_from_columns(const uint64_t *cols, uint32_t leaf_len, uint64_t n_leaves, uint64_t col_stride,
                                    uint32_t cap_height, uint64_t *digests, uint64_t *cap, hipStream_t stream, uint64_t *rows) {
    int lg = log2_exact(n_leaves);
    if (lg < 0 || (int)cap_height > lg) return hipErrorInvalidValue;
    uint32_t log_sub = lg - cap_height;
    hipLaunchKernelGGL(hash_leaves_kernel, dim3(grid_for(n_leaves, 256)), dim3(256), 0, stream, cols, leaf_len, n_leaves,
                       col_stride, digests, cap, log_sub, rows);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    return tree_layers(digests, cap, n_leaves, log_sub, stream);
}

hipError_t hash_leaves_chunk(const uint64_t *cols, uint32_t c0, uint32_t c1, uint32_t leaf_len, uint64_t n_leaves, uint64_t col_stride,
                             uint32_t cap_height, uint64_t *digests, uint64_t *cap, hipStream_t stream, uint64_t *rows, uint64_t rows_from) {
    int lg = log2_exact(n_leaves);
    if (lg < 0 || (int)cap_height > lg || leaf_len <= 4 || (c0 & 7) || c0 >= c1 || c1 > leaf_len) return hipErrorInvalidValue;
    if (c1 != leaf_len && ((c1 - c0) & 7)) return hipErrorInvalidValue;       // inner chunks are whole rate blocks
    if (c1 == leaf_len && (leaf_len & 7) && c1 - c0 < 8 && c0 != 0) return hipErrorInvalidValue;  // see the kernel
    hipLaunchKernelGGL(hash_leaves_chunk_kernel, dim3(grid_for(n_leaves, 256)), dim3(256), 0, stream, cols, c0, c1, leaf_len, n_leaves,
                       col_stride, digests, cap, (uint32_t)(lg - cap_height), rows, rows_from);
    return hipGetLastError();
}

hipError_t merkle_tree_layers(uint64_t *digests, uint64_t *cap, uint64_t n_leaves, uint32_t cap_height, hipStream_t stream) {
    int lg = log2_exact(n_leaves);
    if (lg < 0 || (int)cap_height > lg) return hipErrorInvalidValue;
    return tree_layers(digests, cap, n_leaves, (uint32_t)(lg - cap_height), stream);
}

hipError_t merkle_tree_from_rows(const uint64_t *rows, uint32_t leaf_len, uint64_t n_leaves, uint32_t cap_height,
                                 uint64_t *digests, uint64_t *cap, hipStream_t stream) {
    int lg = log2_exact(n_leaves);
    if (lg < 0 || (int)cap_height > lg) return hipErrorInvalidValue;
    uint32_t log_sub = lg - cap_height;
    if (leaf_len > 4 && n_leaves <= COOP_LAYER_MAX_NODES) {
        poseidon_coop::Tables tb = {};
        hipError_t te = coop_tables(&tb, stream);
        if (te != hipSuccess) return te;
        hipLaunchKernelGGL(hash_rows_coop_kernel, dim3((unsigned)n_leaves), dim3(64), 0, stream, rows, leaf_len, digests, cap, log_sub, tb);
    } else if (n_leaves <= LATENCY_LAYER_MAX_NODES) {
        hipLaunchKernelGGL(hash_rows_kernel<true>, dim3(grid_for(n_leaves, 256)), dim3(256), 0, stream, rows, leaf_len, n_leaves, digests, cap, log_sub);
    } else {
        hipLaunchKernelGGL(hash_rows_kernel<false>, dim3(grid_for(n_leaves, 256)), dim3(256), 0, stream, rows, leaf_len, n_leaves, digests, cap, log_sub);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    return tree_layers(digests, cap, n_leaves, log_sub, stream);
}

hipError_t poseidon_permute_batch(uint64_t *states, uint64_t count, hipStream_t stream) {
    if (count == 0) return hipSuccess;
#ifdef PLONKY2_DEBUG_KNOBS
    static const bool vector_alu = [] {
        const char *e = PLONKY2_KNOB("PLONKY2_POSEIDON");
        return e && e[0] == 'v';
    }();
    if (vector_alu) {
        hipLaunchKernelGGL(permute_batch_vector_kernel, dim3(grid_for(count, 256)), dim3(256), 0, stream, states, count);
        return hipGetLastError();
    }
#endif
    hipLaunchKernelGGL(permute_batch_kernel, dim3(grid_for(count, 256)), dim3(256), 0, stream, states, count);
    return hipGetLastError();
}

hipError_t sponge_absorb(uint64_t *d_state, const uint64_t *d_inputs, uint32_t n_blocks, hipStream_t stream) {
    poseidon_coop::Tables tb;
    hipError_t e = coop_tables(&tb, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(sponge_absorb_kernel, dim3(1), dim3(64), 0, stream, d_state, d_inputs, n_blocks, tb);
    return hipGetLastError();
}

hipError_t merkle_open_batch(const uint64_t *leaves, uint64_t row_stride, uint64_t elem_stride, uint32_t leaf_len, uint64_t n_leaves,
                             uint32_t cap_height, const uint64_t *digests, const uint64_t *d_idx, uint32_t count, uint64_t *out_leaves,
                             uint64_t *out_sib, hipStream_t stream, uint64_t idx_mask, uint32_t idx_shift) {
    if (count == 0) return hipSuccess;
    uint32_t lg = 0;
    while ((1ull << lg) < n_leaves) lg++;
    const uint32_t num_layers = lg - cap_height;
    const uint64_t subtree_digests = 2 * ((n_leaves >> cap_height) - 1);
    hipLaunchKernelGGL(merkle_open_kernel, dim3(count), dim3(64), 0, stream, leaves, row_stride, elem_stride, leaf_len, digests, num_layers,
                       subtree_digests, d_idx, idx_mask, idx_shift, out_leaves, out_sib);
    return hipGetLastError();
}

hipError_t challenger_step(uint64_t *d_challenger, const uint64_t *const *src_ptrs, const uint64_t *src_counts, const uint64_t *src_planar,
                           uint32_t n_src, uint32_t n_out, uint32_t flags, uint64_t *d_out, hipStream_t stream) {
    if (n_src > 8) return hipErrorInvalidValue;
    ChallengerArgs a = {};
    for (uint32_t i = 0; i < n_src; i++) a.src[i] = ChallengerSrc{src_ptrs[i], src_counts[i], src_planar ? src_planar[i] : 0};
    a.n_src = n_src, a.n_out = n_out, a.flags = flags;
    poseidon_coop::Tables tb;
    hipError_t e = coop_tables(&tb, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(challenger_step_kernel, dim3(1), dim3(64), 0, stream, d_challenger, a, d_out, tb);
    return hipGetLastError();
}

// PLONKY2_TRANSPOSE=tile keeps the 64 x 64 tiles (A/B measurements)
static bool strips_enabled() {
    static const bool v = [] {
        const char *e = PLONKY2_KNOB("PLONKY2_TRANSPOSE");
        return !(e && e[0] == 't');
    }();
    return v;
}

hipError_t transpose_to_column_major(const uint64_t *rows, uint64_t *cols, uint32_t n_cols, uint64_t n_rows, uint64_t col_stride,
                                     hipStream_t stream) {
    if (n_cols == 0 || n_rows == 0) return hipSuccess;
    if (strips_enabled()) {
        const StripGeom g = strip_geom(n_cols);
        if (strip_rows32(n_cols))
            hipLaunchKernelGGL(transpose_strip_back_kernel<32>, dim3(grid_for(n_rows, 32), g.n_chunks), dim3(256), (size_t)32 * g.pitch * 8, stream, rows, cols,
                               n_rows, col_stride, g);
        else
            hipLaunchKernelGGL(transpose_strip_back_kernel<64>, dim3(grid_for(n_rows, 64), g.n_chunks), dim3(256), (size_t)64 * g.pitch * 8, stream, rows, cols,
                               n_rows, col_stride, g);
        return hipGetLastError();
    }
    dim3 grid(grid_for(n_rows, TP), grid_for(n_cols, TP));
    hipLaunchKernelGGL(transpose_back_kernel, grid, dim3(256), 0, stream, rows, cols, n_cols, n_rows, col_stride);
    return hipGetLastError();
}

hipError_t transpose_to_leaf_major(const uint64_t *cols, uint64_t *rows, uint32_t n_cols, uint64_t n_rows,
                                   uint64_t col_stride, hipStream_t stream) {
    if (n_cols == 0 || n_rows == 0) return hipSuccess;
    if (strips_enabled()) {
        const StripGeom g = strip_geom(n_cols);
        if (strip_rows32(n_cols))
            hipLaunchKernelGGL(transpose_strip_kernel<32>, dim3(grid_for(n_rows, 32), g.n_chunks), dim3(256), (size_t)32 * g.pitch * 8, stream, cols, rows, n_rows,
                               col_stride, g);
        else
            hipLaunchKernelGGL(transpose_strip_kernel<64>, dim3(grid_for(n_rows, 64), g.n_chunks), dim3(256), (size_t)64 * g.pitch * 8, stream, cols, rows, n_rows,
                               col_stride, g);
        return hipGetLastError();
    }
    dim3 grid(grid_for(n_rows, TP), grid_for(n_cols, TP));
    hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, stream, cols, rows, n_cols, n_rows, col_stride);
    return hipGetLastError();
}

}  // namespace plonky2_hip
