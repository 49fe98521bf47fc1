// plonky2_hip.hpp — header-only C++ mirror of the reference's operator interface for the hot path,
// over the C ABI of plonky2_hip.h. Same names and argument meaning as the Rust side:
//   PolynomialBatch::from_values / from_coeffs / get_lde_values   (plonky2/src/fri/oracle.rs:709-731, 911-1018)
//   MerkleTree::new_ / prove, MerkleCap                            (plonky2/src/hash/merkle_tree.rs:283-319, 392-440)
//   fft_with_options / ifft_with_options                           (field/src/fft.rs:58-103)
// Host containers are std::vector<uint64_t>; everything else stays in HBM. Errors (the reference
// panics or drops them) become plonky2_hip::Error exceptions. No CPU fallback exists.
#pragma once
#include <cstdint>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "plonky2_hip.h"

namespace plonky2_hip {

constexpr uint64_t ORDER = 0xFFFFFFFF00000001ULL;  // goldilocks_field.rs:133
constexpr uint64_t COSET_SHIFT = 7;                // Field::coset_shift, types.rs:431-433
constexpr uint32_t SALT_SIZE = 4;                  // fri/oracle.rs:41

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string &m) : std::runtime_error("plonky2_hip error " + std::to_string(c) + ": " + m), code(c) {}
};

inline void check(GlError e) {
    if (e.code == 0) return;
    std::string msg = e.message ? e.message : cudaGetErrorString(e.code);
    if (e.message) std::free(e.message);
    throw Error(e.code, msg);
}

// Two HIP streams on one device (CudaInnerContext, fri/oracle.rs:43-47).
class Context {
  public:
    explicit Context(int device = 0) : ptr_(gl_ctx_create(device)) {
        if (!ptr_) throw Error(GL_E_INVALID, "gl_ctx_create failed (no HIP device?)");
    }
    ~Context() { gl_ctx_destroy(ptr_); }
    Context(const Context &) = delete;
    Context &operator=(const Context &) = delete;
    void *get() const { return ptr_; }
    void synchronize() const { check(gl_ctx_synchronize(ptr_)); }

  private:
    void *ptr_;
};

// n u64 field elements in HBM.
class DeviceBuffer {
  public:
    DeviceBuffer() = default;
    DeviceBuffer(const Context &ctx, uint64_t n) : ctx_(&ctx), n_(n) {
        void *p = nullptr;
        check(gl_ctx_malloc(&p, n * 8, ctx.get()));
        ptr_ = static_cast<uint64_t *>(p);
    }
    DeviceBuffer(const Context &ctx, const std::vector<uint64_t> &host) : DeviceBuffer(ctx, host.size()) { upload(host); }
    DeviceBuffer(DeviceBuffer &&o) noexcept { *this = std::move(o); }
    DeviceBuffer &operator=(DeviceBuffer &&o) noexcept {
        release();
        ctx_ = o.ctx_; ptr_ = o.ptr_; n_ = o.n_;
        o.ptr_ = nullptr; o.n_ = 0;
        return *this;
    }
    ~DeviceBuffer() { release(); }
    uint64_t *data() const { return ptr_; }
    uint64_t size() const { return n_; }
    void upload(const std::vector<uint64_t> &host, uint64_t offset = 0) {
        if (offset + host.size() > n_) throw Error(GL_E_INVALID, "upload out of range");
        if (!host.empty()) check(gl_memcpy_h2d(ptr_ + offset, host.data(), host.size() * 8, ctx_->get()));
    }
    std::vector<uint64_t> download(uint64_t offset, uint64_t count) const {
        if (offset + count > n_) throw Error(GL_E_INVALID, "download out of range");
        std::vector<uint64_t> out(count);
        if (count) check(gl_memcpy_d2h(out.data(), ptr_ + offset, count * 8, ctx_->get()));
        return out;
    }
    std::vector<uint64_t> download() const { return download(0, n_); }

  private:
    void release() {
        if (ptr_) gl_free(ptr_);
        ptr_ = nullptr;
    }
    const Context *ctx_ = nullptr;
    uint64_t *ptr_ = nullptr;
    uint64_t n_ = 0;
};

inline uint32_t log2_strict(uint64_t n) {
    uint32_t l = 0;
    while ((1ull << l) < n) l++;
    if ((1ull << l) != n) throw Error(GL_E_INVALID, "not a power of two");  // util/src/lib.rs log2_strict panics
    return l;
}

// fft_with_options(poly, zero_factor, root_table): the last two are performance hints in the
// reference (fft.rs:203-217) and are not needed here. polys: n_polys rows of length n.
inline std::vector<uint64_t> fft_with_options(const Context &ctx, const std::vector<uint64_t> &polys, uint64_t n_polys) {
    uint64_t n = polys.size() / n_polys;
    DeviceBuffer d(ctx, polys);
    check(gl_ntt_batch(d.data(), n_polys, log2_strict(n), n, 0, 0, ctx.get()));
    return d.download();
}

inline std::vector<uint64_t> ifft_with_options(const Context &ctx, const std::vector<uint64_t> &polys, uint64_t n_polys) {
    uint64_t n = polys.size() / n_polys;
    DeviceBuffer d(ctx, polys);
    check(gl_ntt_batch(d.data(), n_polys, log2_strict(n), n, 1, 0, ctx.get()));
    return d.download();
}

using HashOut = std::vector<uint64_t>;  // 4 elements (hash_types.rs:17-22)

struct MerkleProof {
    std::vector<HashOut> siblings;  // merkle_proofs.rs:18-22
};

class MerkleTree {
  public:
    // MerkleTree::new(leaves, cap_height) (merkle_tree.rs:283-319); leaves leaf-major [n_leaves][leaf_len].
    static MerkleTree new_(const Context &ctx, const std::vector<uint64_t> &leaves, uint64_t n_leaves, uint32_t cap_height) {
        uint32_t lg = log2_strict(n_leaves);
        if (cap_height > lg)
            throw Error(GL_E_INVALID, "cap_height=" + std::to_string(cap_height) + " should be at most log2(leaves.len())=" + std::to_string(lg));
        MerkleTree t;
        t.ctx_ = &ctx;
        t.n_leaves_ = n_leaves;
        t.leaf_len_ = (uint32_t)(leaves.size() / n_leaves);
        t.cap_height_ = cap_height;
        t.d_leaves_ = DeviceBuffer(ctx, leaves);
        t.d_digests_ = DeviceBuffer(ctx, 8 * (n_leaves - (1ull << cap_height)) + 4);
        t.d_cap_ = DeviceBuffer(ctx, 4ull << cap_height);
        check(gl_merkle_tree_from_leaves(t.d_leaves_.data(), t.leaf_len_, n_leaves, cap_height, t.d_digests_.data(),
                                         t.d_cap_.data(), ctx.get()));
        return t;
    }
    // adopt buffers produced by a commit
    static MerkleTree adopt(const Context &ctx, uint64_t n_leaves, uint32_t leaf_len, uint32_t cap_height, DeviceBuffer digests,
                            DeviceBuffer cap, DeviceBuffer leaves) {
        MerkleTree t;
        t.ctx_ = &ctx; t.n_leaves_ = n_leaves; t.leaf_len_ = leaf_len; t.cap_height_ = cap_height;
        t.d_digests_ = std::move(digests); t.d_cap_ = std::move(cap); t.d_leaves_ = std::move(leaves);
        return t;
    }
    uint64_t num_digests() const { return 2 * (n_leaves_ - (1ull << cap_height_)); }
    std::vector<uint64_t> cap() const { return d_cap_.download(0, 4ull << cap_height_); }
    std::vector<uint64_t> digests() const { return d_digests_.download(0, 4 * num_digests()); }
    std::vector<uint64_t> get(uint64_t i) const { return d_leaves_.download(i * leaf_len_, leaf_len_); }  // merkle_tree.rs:385-391
    // MerkleTree::prove (merkle_tree.rs:392-440)
    MerkleProof prove(uint64_t leaf_index) const {
        uint32_t num_layers = log2_strict(n_leaves_) - cap_height_;
        uint64_t tree_len = num_digests() >> cap_height_;
        uint64_t tree_index = leaf_index >> num_layers;
        uint64_t pair_index = leaf_index & ((1ull << num_layers) - 1);
        MerkleProof p;
        for (uint32_t i = 0; i < num_layers; i++) {
            uint64_t parity = pair_index & 1;
            pair_index >>= 1;
            uint64_t siblings_index = (pair_index << (i + 1)) + (1ull << i) - 1;
            uint64_t sibling_index = 2 * siblings_index + (1 - parity);
            p.siblings.push_back(d_digests_.download(4 * (tree_len * tree_index + sibling_index), 4));
        }
        return p;
    }
    uint64_t n_leaves() const { return n_leaves_; }
    uint32_t leaf_len() const { return leaf_len_; }

  private:
    const Context *ctx_ = nullptr;
    uint64_t n_leaves_ = 0;
    uint32_t leaf_len_ = 0, cap_height_ = 0;
    DeviceBuffer d_leaves_, d_digests_, d_cap_;
};

// PolynomialBatch (fri/oracle.rs:112-120): coefficients, LDE and tree stay resident in HBM.
class PolynomialBatch {
  public:
    // from_values(values, rate_bits, blinding, cap_height, timing, fft_root_table) (oracle.rs:709-731).
    // values: n_polys columns of length n, column-major. `salt` = SALT_SIZE columns of n<<rate_bits
    // elements when blinding (the reference draws them from OsRng, oracle.rs:998-1002).
    static PolynomialBatch from_values(const Context &ctx, const std::vector<uint64_t> &values, uint64_t n_polys, uint32_t rate_bits,
                                       bool blinding, uint32_t cap_height, const std::vector<uint64_t> &salt = {}) {
        return commit(ctx, values, n_polys, rate_bits, blinding, cap_height, salt, true);
    }
    // from_coeffs (oracle.rs:911-977)
    static PolynomialBatch from_coeffs(const Context &ctx, const std::vector<uint64_t> &coeffs, uint64_t n_polys, uint32_t rate_bits,
                                       bool blinding, uint32_t cap_height, const std::vector<uint64_t> &salt = {}) {
        return commit(ctx, coeffs, n_polys, rate_bits, blinding, cap_height, salt, false);
    }
    std::vector<uint64_t> polynomials() const { return d_polys_.download(); }
    // get_lde_values(index, step) (oracle.rs:1007-1018)
    std::vector<uint64_t> get_lde_values(uint64_t index, uint64_t step = 1) const {
        uint64_t idx = index * step, bits = degree_log + rate_bits, rev = 0;
        for (uint64_t b = 0; b < bits; b++) rev |= ((idx >> b) & 1) << (bits - 1 - b);
        std::vector<uint64_t> row = merkle_tree.get(rev);
        row.resize(row.size() - (blinding ? SALT_SIZE : 0));
        return row;
    }
    MerkleTree merkle_tree;
    uint32_t degree_log = 0, rate_bits = 0;
    bool blinding = false;

  private:
    static PolynomialBatch commit(const Context &ctx, const std::vector<uint64_t> &polys, uint64_t n_polys, uint32_t rate_bits,
                                  bool blinding, uint32_t cap_height, const std::vector<uint64_t> &salt, bool is_values) {
        uint64_t n = polys.size() / n_polys;
        uint32_t lg = log2_strict(n);
        uint64_t n_ext = n << rate_bits;
        uint32_t salt_size = blinding ? SALT_SIZE : 0;
        if (cap_height > lg + rate_bits) throw Error(GL_E_INVALID, "cap_height should be at most log2(leaves.len())");
        if (salt.size() != (uint64_t)salt_size * n_ext) throw Error(GL_E_INVALID, "blinding needs SALT_SIZE columns of salt");
        uint64_t cols = n_polys + salt_size;
        PolynomialBatch b;
        b.degree_log = lg; b.rate_bits = rate_bits; b.blinding = blinding;
        b.d_polys_ = DeviceBuffer(ctx, polys);
        b.d_lde_ = DeviceBuffer(ctx, cols * n_ext);
        if (salt_size) b.d_lde_.upload(salt, n_polys * n_ext);
        DeviceBuffer leaves(ctx, cols * n_ext), dig(ctx, 8 * (n_ext - (1ull << cap_height)) + 4), cap(ctx, 4ull << cap_height);
        check((is_values ? gl_commit_from_values : gl_commit_from_coeffs_nc)(b.d_polys_.data(), n_polys, lg, rate_bits, cap_height, salt_size,
                                                                              COSET_SHIFT, b.d_lde_.data(), leaves.data(), dig.data(),
                                                                              cap.data(), ctx.get()));
        ctx.synchronize();
        b.merkle_tree = MerkleTree::adopt(ctx, n_ext, (uint32_t)cols, cap_height, std::move(dig), std::move(cap), std::move(leaves));
        return b;
    }
    static GlError gl_commit_from_coeffs_nc(uint64_t *c, uint64_t p, uint32_t l, uint32_t r, uint32_t h, uint32_t s, uint64_t sh, uint64_t *lde,
                                            uint64_t *lv, uint64_t *dg, uint64_t *cp, void *ctx) {
        return gl_commit_from_coeffs(c, p, l, r, h, s, sh, lde, lv, dg, cp, ctx);
    }
    DeviceBuffer d_polys_, d_lde_;
};

}  // namespace plonky2_hip
