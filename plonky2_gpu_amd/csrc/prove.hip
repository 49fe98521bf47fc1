// prove.hip — the prover's HOST logic in native code: gl_circuit_create / gl_prove.
//
// prove() (plonky2/src/plonk/prover.rs:40-233) from the full witness on, with PolynomialBatch::prove_openings
// (plonky2/src/fri/oracle.rs:1047-1112), fri_proof (plonky2/src/fri/prover.rs:24-260), the Challenger
// (plonky2/src/iop/challenger.rs) and the proof wire format (plonky2/src/util/serialization.rs:466-700).
// Everything data-parallel is a kernel behind the gl_* entry points of this library; what lives here is
// the serial glue a Rust host would otherwise write against those entry points — the transcript's
// buffers, challenge arithmetic on single field elements, buffer management, serialisation — so that a
// host in any language needs exactly two calls. No polynomial, LDE, tree or witness column is touched
// by the CPU; the transcript lives on the device (gl_challenger_step): its sponge state, input buffer and challenges.
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/plonky2_hip.h"
#include "gate_jit.h"
#include "gl_field.h"

namespace {

using glh::P;

#define TRY(expr)                  \
    do {                           \
        GlError _e = (expr);       \
        if (_e.code != 0) return _e; \
    } while (0)

GlError ok() { return GlError{0, nullptr}; }
GlError fail(const std::string &m) { return GlError{GL_E_INVALID, strdup(m.c_str())}; }

struct E2 {  // a + bX, X^2 = 7 (field/src/goldilocks_extensions.rs:13-26)
    uint64_t a, b;
};
E2 e2_mul(E2 x, E2 y) {
    return E2{glh::add(glh::mul(x.a, y.a), glh::mul(7, glh::mul(x.b, y.b))), glh::add(glh::mul(x.a, y.b), glh::mul(x.b, y.a))};
}
E2 e2_pow(E2 x, uint64_t e) {
    E2 acc{1, 0};
    while (e) {
        if (e & 1) acc = e2_mul(acc, x);
        x = e2_mul(x, x);
        e >>= 1;
    }
    return acc;
}

// Device buffers of one circuit's proofs are recycled: every proof of a circuit allocates the same ~50
// sizes, hipMalloc/hipFree of multi-GiB buffers cost milliseconds and hipFree synchronises the device.
// Handing a buffer back while kernels that use it are still in flight is safe here because everything
// gl_prove launches is ordered on ONE stream: the next user's kernels queue behind them. That argument
// holds per CONTEXT, so a circuit keeps one pool per context that proves with it (two proofs of one
// circuit in flight on two contexts never hand each other a buffer whose kernels are still queued).
struct Pool {
    std::mutex m;
    std::multimap<uint64_t, uint64_t *> free_;  // bytes -> buffer
    // page-locked host staging of this context's proofs: what the host sends (circuit digest, public inputs) and everything it
    // fetches (challenges, caps, openings, query answers) goes through it, so that the copies are truly asynchronous
    uint64_t *pinned = nullptr;
    uint64_t pinned_words = 0;
    ~Pool() {
        for (auto &kv : free_) (void)gl_free(kv.second);
        if (pinned) (void)gl_free_host(pinned);
    }
    GlError staging(uint64_t words, uint64_t **out) {
        if (pinned_words < words) {
            if (pinned) (void)gl_free_host(pinned);
            pinned = nullptr, pinned_words = 0;
            void *q = nullptr;
            GlError e = gl_malloc_host(&q, words * 8);
            if (e.code != 0) return e;
            pinned = static_cast<uint64_t *>(q), pinned_words = words;
        }
        *out = pinned;
        return GlError{0, nullptr};
    }
    uint64_t *get(uint64_t bytes) {
        std::lock_guard<std::mutex> lock(m);
        auto it = free_.find(bytes);
        if (it == free_.end()) return nullptr;
        uint64_t *p = it->second;
        free_.erase(it);
        return p;
    }
    void put(uint64_t bytes, uint64_t *p) {
        std::lock_guard<std::mutex> lock(m);
        free_.emplace(bytes, p);
    }
};
thread_local Pool *g_pool = nullptr;  // installed by gl_prove for its duration; null = plain gl_malloc / gl_free
struct PoolScope {
    Pool *prev;
    explicit PoolScope(Pool *p) : prev(g_pool) { g_pool = p; }
    ~PoolScope() { g_pool = prev; }
};

struct DevBuf {  // RAII device buffer of u64
    uint64_t *p = nullptr;
    uint64_t n = 0, bytes = 0;
    Pool *pool = nullptr;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    DevBuf(DevBuf &&o) noexcept : p(o.p), n(o.n), bytes(o.bytes), pool(o.pool) { o.p = nullptr; }
    DevBuf &operator=(DevBuf &&o) noexcept {
        reset();
        p = o.p, n = o.n, bytes = o.bytes, pool = o.pool, o.p = nullptr;
        return *this;
    }
    ~DevBuf() { reset(); }
    void reset() {
        if (p) {
            if (pool)
                pool->put(bytes, p);
            else
                (void)gl_free(p);
        }
        p = nullptr;
    }
    GlError alloc(uint64_t elems) {
        reset();
        n = elems;
        bytes = (elems ? elems : 1) * 8;
        pool = g_pool;
        if (pool && (p = pool->get(bytes))) return ok();
        void *q = nullptr;
        TRY(gl_malloc(&q, bytes));
        p = static_cast<uint64_t *>(q);
        return ok();
    }
};

// A committed batch resident in HBM (PolynomialBatch, fri/oracle.rs:112-120), without a leaf-major copy.
struct Batch {
    DevBuf coeffs, lde, digests, cap_d;
    uint32_t n_polys = 0;
    uint32_t leaf_len = 0;  // n_polys + the salt of a blinded commitment (fri/oracle.rs:985-1002)
    std::vector<uint64_t> cap;  // host copy, 4 << cap_height
};

// ---- transcript: device-resident (gl_challenger_step), see prove_impl; gl_circuit_create hashes a few host words once ----
GlError hash_no_pad(const uint64_t *in, size_t n, uint64_t out[4], void *ctx) {  // hash/hashing.rs:81-108
    uint64_t st[12] = {0};
    std::vector<uint64_t> v(n);
    for (size_t i = 0; i < n; i++) v[i] = in[i] % P;
    const size_t full = n / 8 * 8;
    if (full) TRY(gl_sponge_absorb(st, v.data(), (uint32_t)(full / 8), ctx));
    if (full < n) {  // a short last chunk leaves the old lanes in place
        for (size_t i = full; i < n; i++) st[i - full] = v[i];
        uint64_t block[8];
        memcpy(block, st, sizeof block);
        TRY(gl_sponge_absorb(st, block, 1, ctx));
    }
    memcpy(out, st, 32);
    return ok();
}

// ---- the circuit object ---------------------------------------------------------------------------
struct Circuit {
    uint32_t degree_bits, num_wires, num_routed, num_constants, num_challenges, qdf, num_gate_constraints;
    uint32_t rate_bits, cap_height, pow_bits, num_queries;
    bool hiding = false;  // FriParams::hiding
    std::vector<uint32_t> arity_bits;
    uint64_t digest[4];
    DevBuf k_is, sigmas;
    Batch cs;  // constants_sigmas_commitment
    // gates
    DevBuf d_instrs, d_gates, d_imms;
    uint32_t num_gates = 0, num_selectors = 0;
    void *gate_kernel = nullptr;
    mutable std::mutex pools_m;
    mutable std::map<void *, Pool> pools;  // context -> the working buffers of this circuit's proofs there, recycled from proof to proof
    Pool *pool_of(void *ctx) const {
        std::lock_guard<std::mutex> lock(pools_m);
        return &pools[ctx];  // std::map: the address is stable
    }
    ~Circuit() {
        if (gate_kernel) gl_gate_kernel_destroy(gate_kernel);
    }
};

uint32_t num_partial_products(uint32_t routed, uint32_t qdf) { return (routed + qdf - 1) / qdf - 1; }

constexpr uint32_t SALT_SIZE = 4;  // fri/oracle.rs:41

__global__ void canon_copy_kernel(uint64_t *dst, const uint64_t *src, uint64_t n) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) dst[i] = gl::canon(src[i]);
}
GlError canon_copy(uint64_t *d_dst, const uint64_t *d_src, uint64_t n, void *ctx) {
    if (n == 0) return ok();
    const uint64_t blocks = (n + 255) / 256;
    hipLaunchKernelGGL(canon_copy_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, *reinterpret_cast<hipStream_t *>(ctx), d_dst, d_src, n);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(hipGetErrorString(e));
    return ok();
}

// d_salt: SALT_SIZE columns of n_ext caller-provided random elements in leaf order (a blinded commitment, prover.rs:84, 125, 174), or null
GlError commit(Batch *b, DevBuf &&polys, bool from_values, uint32_t n_polys, const Circuit &c, void *ctx, const uint64_t *d_salt = nullptr,
              bool fetch_cap = true) {
    const uint64_t n_ext = 1ull << (c.degree_bits + c.rate_bits);
    const uint32_t salt = d_salt ? SALT_SIZE : 0;
    b->coeffs = std::move(polys);
    b->n_polys = n_polys;
    b->leaf_len = n_polys + salt;
    TRY(b->lde.alloc((uint64_t)b->leaf_len * n_ext));
    TRY(b->digests.alloc(8 * (n_ext - (1ull << c.cap_height))));
    TRY(b->cap_d.alloc(4ull << c.cap_height));
    // the salt columns sit behind the LDE's columns and are hashed with them (gl_commit_from_* reads them as given). They also go into
    // the proof verbatim (fri/prover.rs:203-210), where every word must be canonical like the reference's F::rand_vec output
    // (fri/oracle.rs:998-1002): a caller who fills d_salts with raw 64-bit randoms gets them reduced here, not >= p words on the wire.
    if (salt) TRY(canon_copy(b->lde.p + (uint64_t)n_polys * n_ext, d_salt, (uint64_t)salt * n_ext, ctx));
    if (from_values)
        TRY(gl_commit_from_values(b->coeffs.p, n_polys, c.degree_bits, c.rate_bits, c.cap_height, salt, 7, b->lde.p, nullptr, b->digests.p,
                                  b->cap_d.p, ctx));
    else
        TRY(gl_commit_from_coeffs(b->coeffs.p, n_polys, c.degree_bits, c.rate_bits, c.cap_height, salt, 7, b->lde.p, nullptr, b->digests.p,
                                  b->cap_d.p, ctx));
    b->cap.resize(4ull << c.cap_height);
    if (!fetch_cap) return ok();  // gl_prove: the transcript reads the cap where it lies; the host copy is fetched with the rest of the proof
    return gl_memcpy_d2h(b->cap.data(), b->cap_d.p, b->cap.size() * 8, ctx);
}

struct Bytes {  // util/serialization.rs:466-560
    std::vector<uint8_t> v;
    void u8(uint8_t x) { v.push_back(x); }
    void field(uint64_t x) {
        x %= P;
        for (int i = 0; i < 8; i++) v.push_back((uint8_t)(x >> (8 * i)));
    }
    void fields(const uint64_t *p, size_t n) {
        for (size_t i = 0; i < n; i++) field(p[i]);
    }
    void fields(const std::vector<uint64_t> &a) { fields(a.data(), a.size()); }
    void merkle_proof(const uint64_t *sib, uint32_t layers) {  // :573-589
        u8((uint8_t)layers);
        fields(sib, 4ull * layers);
    }
};

struct Stages {
    double *ms;
    void *ctx;
    std::chrono::steady_clock::time_point t;
    Stages(double *m, void *c) : ms(m), ctx(c), t(std::chrono::steady_clock::now()) {}
    GlError mark(int i) {
        if (!ms) return ok();
        TRY(gl_ctx_synchronize(ctx));
        auto now = std::chrono::steady_clock::now();
        ms[i] += std::chrono::duration<double, std::milli>(now - t).count();
        t = now;
        return ok();
    }
};

}  // namespace

extern "C" {

GlError gl_circuit_create(const GlCircuitDesc *d, void **circuit, void *ctx) {
    if (!d || !circuit || !ctx || !d->h_k_is || !d->h_constants || !d->h_sigmas || (d->fri.num_reductions && !d->fri.reduction_arity_bits))
        return fail("null pointer");
    if (d->struct_size != sizeof(GlCircuitDesc))
        return fail("GlCircuitDesc.struct_size does not equal sizeof(GlCircuitDesc) of this library: the caller was compiled against another version of include/plonky2_hip.h");
    if (d->degree_bits > 24 || d->num_challenges == 0 || d->num_challenges > 4 || d->num_routed_wires > d->num_wires ||
        d->quotient_degree_factor < 2 || d->quotient_degree_factor >= d->num_routed_wires)
        return fail("bad circuit shape (the prover needs quotient_degree_factor < num_routed_wires, prover.rs:99-102)");
    Circuit *c = new Circuit();
    c->degree_bits = d->degree_bits, c->num_wires = d->num_wires, c->num_routed = d->num_routed_wires;
    c->num_constants = d->num_constants, c->num_challenges = d->num_challenges, c->qdf = d->quotient_degree_factor;
    c->num_gate_constraints = d->num_gate_constraints;
    c->rate_bits = d->fri.rate_bits, c->cap_height = d->fri.cap_height, c->pow_bits = d->fri.proof_of_work_bits;
    c->num_queries = d->fri.num_query_rounds;
    c->hiding = d->fri.hiding != 0;
    c->arity_bits.assign(d->fri.reduction_arity_bits, d->fri.reduction_arity_bits + d->fri.num_reductions);
    const uint64_t n = 1ull << c->degree_bits;
    auto bail = [&](GlError e) {
        delete c;
        return e;
    };
#define CTRY(expr)                         \
    do {                                   \
        GlError _e = (expr);               \
        if (_e.code != 0) return bail(_e); \
    } while (0)
    CTRY(c->k_is.alloc(c->num_routed));
    CTRY(gl_memcpy_h2d(c->k_is.p, d->h_k_is, 8ull * c->num_routed, ctx));
    CTRY(c->sigmas.alloc((uint64_t)c->num_routed * n));
    CTRY(gl_memcpy_h2d(c->sigmas.p, d->h_sigmas, 8ull * c->num_routed * n, ctx));
    // constants_sigmas_commitment (circuit_builder.rs:861-873): constants then sigmas, from values
    DevBuf csv;
    CTRY(csv.alloc((uint64_t)(c->num_constants + c->num_routed) * n));
    CTRY(gl_memcpy_h2d(csv.p, d->h_constants, 8ull * c->num_constants * n, ctx));
    CTRY(gl_memcpy_h2d(csv.p + (uint64_t)c->num_constants * n, d->h_sigmas, 8ull * c->num_routed * n, ctx));
    CTRY(commit(&c->cs, std::move(csv), true, c->num_constants + c->num_routed, *c, ctx));
    if (d->h_circuit_digest) {
        memcpy(c->digest, d->h_circuit_digest, 32);
    } else {
        // circuit_builder.rs:915-927: hash_no_pad(cap || hash_pad(domain separator = []) || degree_bits)
        uint64_t pad[12] = {1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1}, dsd[4];
        CTRY(hash_no_pad(pad, 12, dsd, ctx));
        std::vector<uint64_t> parts(c->cs.cap);
        parts.insert(parts.end(), dsd, dsd + 4);
        parts.push_back(c->degree_bits);
        CTRY(hash_no_pad(parts.data(), parts.size(), c->digest, ctx));
    }
    if (d->num_gates) {
        if (!d->h_instrs || !d->h_gates) return bail(fail("null gate program"));
        c->num_gates = d->num_gates, c->num_selectors = d->num_selectors;
        {
            std::string verr;
            uint32_t wires_needed = 0, constants_needed = 0;
            if (!plonky2_hip::gate_programs_validate(reinterpret_cast<const uint16_t *>(d->h_instrs), d->num_instrs, reinterpret_cast<const uint32_t *>(d->h_gates),
                                        d->num_gates, d->num_immediates, d->num_selectors, d->num_gate_constraints, &wires_needed,
                                        &constants_needed, &verr))
                return bail(fail("gate programs: " + verr));
            if (wires_needed > c->num_wires || constants_needed > c->num_constants)
                return bail(fail("gate programs load wire " + std::to_string(wires_needed ? wires_needed - 1 : 0) + " / constant column " +
                                 std::to_string(constants_needed ? constants_needed - 1 : 0) + " but the circuit has " + std::to_string(c->num_wires) +
                                 " wires and " + std::to_string(c->num_constants) + " constants"));
        }
        if (d->compile_gates) {
            CTRY(gl_gate_kernel_build(d->h_instrs, d->num_instrs, d->h_gates, d->num_gates, d->h_immediates, d->num_immediates,
                                      d->num_selectors, d->num_gate_constraints, d->num_challenges, &c->gate_kernel));
        } else {
            CTRY(c->d_instrs.alloc(d->num_instrs ? d->num_instrs : 1));  // 8 bytes per GlGateInstr
            CTRY(gl_memcpy_h2d(c->d_instrs.p, d->h_instrs, 8ull * d->num_instrs, ctx));
            CTRY(c->d_gates.alloc(3ull * d->num_gates));  // 24 bytes per GlGateDesc
            CTRY(gl_memcpy_h2d(c->d_gates.p, d->h_gates, 24ull * d->num_gates, ctx));
            if (d->num_immediates) {
                CTRY(c->d_imms.alloc(d->num_immediates));
                CTRY(gl_memcpy_h2d(c->d_imms.p, d->h_immediates, 8ull * d->num_immediates, ctx));
            }
        }
    }
    CTRY(gl_ctx_synchronize(ctx));
#undef CTRY
    *circuit = c;
    return ok();
}

void gl_circuit_destroy(void *circuit) { delete static_cast<Circuit *>(circuit); }

GlError gl_circuit_trim(void *circuit) {
    if (!circuit) return fail("null pointer");
    Circuit *c = static_cast<Circuit *>(circuit);
    std::lock_guard<std::mutex> all(c->pools_m);
    for (auto &cp : c->pools) {
        Pool &pool = cp.second;
        std::lock_guard<std::mutex> lock(pool.m);
        for (auto &kv : pool.free_) TRY(gl_free(kv.second));
        pool.free_.clear();
    }
    return ok();
}

GlError gl_circuit_info(const void *circuit, uint64_t h_digest[4], uint64_t *h_constants_sigmas_cap) {
    if (!circuit) return fail("null pointer");
    const Circuit *c = static_cast<const Circuit *>(circuit);
    if (h_digest) memcpy(h_digest, c->digest, 32);
    if (h_constants_sigmas_cap) memcpy(h_constants_sigmas_cap, c->cs.cap.data(), c->cs.cap.size() * 8);
    return ok();
}

void gl_bytes_free(uint8_t *p) { free(p); }

// Asynchronous copies between device memory and the pool's page-locked staging, on the context's first stream.
static GlError copy_async(void *dst, const void *src, uint64_t bytes, bool to_host, void *ctx) {
    if (!bytes) return ok();
    const hipError_t e = hipMemcpyAsync(dst, src, bytes, to_host ? hipMemcpyDeviceToHost : hipMemcpyHostToDevice, *reinterpret_cast<hipStream_t *>(ctx));
    if (e != hipSuccess) return fail(std::string("hipMemcpyAsync: ") + hipGetErrorString(e));
    return ok();
}
static GlError stream_sync(void *ctx) {
    const hipError_t e = hipStreamSynchronize(*reinterpret_cast<hipStream_t *>(ctx));
    if (e != hipSuccess) return fail(std::string("hipStreamSynchronize: ") + hipGetErrorString(e));
    return ok();
}

// The transcript of a proof lives on the device (gl_challenger_step): every observation reads its source where the producing kernel
// left it (caps, openings, the final polynomial, the proof-of-work witness), the FRI betas, the query indices and the proof-of-work
// state are consumed there, and the host fetches exactly the challenges it computes with — betas / gammas (with the public-inputs
// hash), alphas, zeta, the FRI alpha — one small copy and one stream synchronisation each, then the proof-of-work witness, then
// everything that goes into the proof bytes in one go. Round 5 paid a host round trip per Challenger call (sixteen per proof, each
// an upload, a launch, a download and a synchronisation) and one per cap, opening batch and query batch.
static GlError prove_impl(const void *circuit, const uint64_t *d_wires, const uint64_t *h_public_inputs, uint32_t num_public_inputs,
                          const uint64_t *d_salts, uint8_t **proof, uint64_t *proof_len, double *h_stage_ms, void *ctx) {
    if (!circuit || !d_wires || !proof || !proof_len || !ctx || (num_public_inputs && !h_public_inputs)) return fail("null pointer");
    const Circuit &c = *static_cast<const Circuit *>(circuit);
    if (c.hiding && !d_salts) return fail("the circuit's FRI parameters are hiding (zero_knowledge): prove it with gl_prove_zk and salt columns");
    if (!c.hiding && d_salts) return fail("gl_prove_zk on a circuit whose FRI parameters are not hiding");
    Pool *pool = c.pool_of(ctx);
    PoolScope pool_scope(pool);  // every DevBuf below comes from / returns to the circuit's pool of this context
    const uint32_t db = c.degree_bits, nch = c.num_challenges, qdf = c.qdf, nq = c.num_queries, npi = num_public_inputs;
    const uint64_t n = 1ull << db, n_ext = n << c.rate_bits, cap_words = 4ull << c.cap_height;
    const uint32_t npp = num_partial_products(c.num_routed, qdf);
    const uint32_t lg_ext = db + c.rate_bits, init_layers = lg_ext - c.cap_height, n_fri = (uint32_t)c.arity_bits.size();
    if (h_stage_ms) memset(h_stage_ms, 0, sizeof(double) * GL_PROVE_STAGES);
    Stages st(h_stage_ms, ctx);

    // ---- the small-data side of the proof: one device buffer, one page-locked mirror -------------------------------------------
    const uint32_t n_polys[4] = {c.num_constants + c.num_routed, c.num_wires, nch * (1 + npp), nch * qdf};
    const uint32_t salt = d_salts ? SALT_SIZE : 0;
    const uint32_t leaf_len[4] = {n_polys[0], n_polys[1] + salt, n_polys[2] + salt, n_polys[3] + salt};
    struct FriShape {
        uint64_t n_leaves;
        uint32_t leaf_len, layers, shift;
    };
    std::vector<FriShape> fs(n_fri);
    uint64_t final_len = n;
    {
        uint64_t len = n;
        uint32_t shift = 0;
        for (uint32_t li = 0; li < n_fri; li++) {
            const uint32_t ab = c.arity_bits[li];
            fs[li].n_leaves = (len << c.rate_bits) >> ab, fs[li].leaf_len = 2u << ab;
            if (fs[li].n_leaves < (1ull << c.cap_height)) return fail("FRI layer smaller than the Merkle cap");
            uint32_t lg = 0;
            while ((1ull << lg) < fs[li].n_leaves) lg++;
            fs[li].layers = lg - c.cap_height;
            shift += ab;
            fs[li].shift = shift;
            len >>= ab;
        }
        final_len = len;
    }
    struct Span {
        uint64_t off = 0, words = 0;
    };
    uint64_t top = 0;
    auto take = [&](uint64_t words) {
        Span sp{top, words};
        top += (words + 1) & ~1ull;  // 16-byte granules
        return sp;
    };
    const Span T = take(32), HP = take(32), hostin = take(4 + (uint64_t)npi);
    const Span fetch0 = take(0);  // from here on: what the host fetches
    const Span pih_s = take(4), bg = take(2ull * nch), alphas_s = take(nch), zeta_s = take(2), alpha_fri_s = take(2), fri_betas = take(2ull * n_fri);
    const Span pow_w = take(1), resp_idx = take(1 + (uint64_t)nq);
    Span opens[4], caps[3], fri_caps = take(cap_words * n_fri), final_s = take(2 * final_len);
    for (int o = 0; o < 4; o++) opens[o] = take(2ull * (o == 2 ? 2 : 1) * n_polys[o]);
    for (int o = 0; o < 3; o++) caps[o] = take(cap_words);
    Span q_leaves[4], q_sib[4];
    for (int o = 0; o < 4; o++) q_leaves[o] = take((uint64_t)nq * leaf_len[o]), q_sib[o] = take((uint64_t)nq * init_layers * 4);
    std::vector<Span> s_leaves(n_fri), s_sib(n_fri);
    for (uint32_t li = 0; li < n_fri; li++) s_leaves[li] = take((uint64_t)nq * fs[li].leaf_len), s_sib[li] = take((uint64_t)nq * fs[li].layers * 4);
    DevBuf small;
    TRY(small.alloc(top));
    uint64_t *const D = small.p;
    uint64_t *H = nullptr;
    TRY(pool->staging(top, &H));
    auto fetch = [&](const Span &sp) { return copy_async(H + sp.off, D + sp.off, sp.words * 8, true, ctx); };
    auto step = [&](std::initializer_list<GlObserveSrc> srcs, uint32_t n_out, const Span &out, uint32_t flags = 0, const Span *which = nullptr) {
        return gl_challenger_step(D + (which ? which->off : T.off), srcs.begin(), (uint32_t)srcs.size(), n_out, n_out || (flags & GL_CHALLENGER_HASH) ? D + out.off : nullptr,
                                  flags, ctx);
    };

    // circuit digest and public inputs go up once; hash_no_pad(public inputs) (prover.rs:52) on a scratch challenger
    memcpy(H + hostin.off, c.digest, 32);
    for (uint32_t i = 0; i < npi; i++) H[hostin.off + 4 + i] = h_public_inputs[i];
    TRY(copy_async(D + hostin.off, H + hostin.off, hostin.words * 8, false, ctx));
    TRY(step({GlObserveSrc{D + hostin.off + 4, npi, 0}}, 0, pih_s, GL_CHALLENGER_RESET | GL_CHALLENGER_HASH, &HP));
    // wires commitment (prover.rs:66-90); the caller's witness stays intact for the partial products
    Batch wires;
    {
        DevBuf w;
        TRY(w.alloc((uint64_t)c.num_wires * n));
        TRY(gl_memcpy_d2d(w.p, d_wires, 8ull * c.num_wires * n, ctx));
        TRY(commit(&wires, std::move(w), true, c.num_wires, c, ctx, d_salts, false));
    }
    TRY(st.mark(0));
    // challenger.observe_hash(circuit digest), observe_hash(public inputs hash), observe_cap(wires cap); betas, gammas (prover.rs:92-97)
    TRY(step({GlObserveSrc{D + hostin.off, 4, 0}, GlObserveSrc{D + pih_s.off, 4, 0}, GlObserveSrc{wires.cap_d.p, cap_words, 0}}, 2 * nch, bg, GL_CHALLENGER_RESET));
    TRY(fetch(Span{pih_s.off, bg.off + bg.words - pih_s.off}));
    TRY(stream_sync(ctx));
    uint64_t pih[4];
    memcpy(pih, H + pih_s.off, 32);
    const std::vector<uint64_t> betas(H + bg.off, H + bg.off + nch), gammas(H + bg.off + nch, H + bg.off + 2 * nch);
    std::vector<uint64_t> alphas;
    // partial products and Z (prover.rs:99-117), committed in place
    Batch zs;
    {
        DevBuf z;
        TRY(z.alloc((uint64_t)nch * (1 + npp) * n));
        TRY(gl_permutation_partial_products(d_wires, n, c.sigmas.p, n, c.k_is.p, betas.data(), gammas.data(), nch, c.num_routed, qdf, db, z.p,
                                            ctx));
        TRY(st.mark(1));
        TRY(commit(&zs, std::move(z), true, nch * (1 + npp), c, ctx, d_salts ? d_salts + (uint64_t)SALT_SIZE * n_ext : nullptr, false));
    }
    TRY(st.mark(2));
    TRY(step({GlObserveSrc{zs.cap_d.p, cap_words, 0}}, nch, alphas_s));
    TRY(fetch(alphas_s));
    TRY(stream_sync(ctx));
    alphas.assign(H + alphas_s.off, H + alphas_s.off + nch);
    // quotient polynomials (prover.rs:137-151)
    uint32_t qdb = 0;
    while ((1u << qdb) < qdf) qdb++;
    DevBuf quotient, work;
    TRY(quotient.alloc((uint64_t)nch << (db + qdb)));
    {
        GlQuotientArgs a;
        memset(&a, 0, sizeof a);
        a.d_wires_leaves = wires.lde.p, a.d_constants_sigmas_leaves = c.cs.lde.p, a.d_zs_partial_products_leaves = zs.lde.p;
        a.wires_leaf_len = c.num_wires, a.constants_sigmas_leaf_len = c.num_constants + c.num_routed;
        a.zs_partial_products_leaf_len = nch * (1 + npp);
        a.d_k_is = c.k_is.p;
        a.h_betas = betas.data(), a.h_gammas = gammas.data(), a.h_alphas = alphas.data();
        a.num_constants = c.num_constants, a.num_routed_wires = c.num_routed, a.num_challenges = nch;
        a.num_gate_constraints = c.num_gates ? c.num_gate_constraints : 0;
        a.degree_bits = db, a.rate_bits = c.rate_bits, a.quotient_degree_factor = qdf, a.coset_shift = 7;
        a.column_stride = n_ext;
        GlGateProgram gp;
        memset(&gp, 0, sizeof gp);
        if (c.gate_kernel) {
            TRY(work.alloc((uint64_t)nch << (db + qdb)));
            a.gate_kernel = c.gate_kernel, a.h_public_inputs_hash = pih, a.d_gate_workspace = work.p;
        } else if (c.num_gates) {
            gp.d_instrs = reinterpret_cast<const GlGateInstr *>(c.d_instrs.p);
            gp.d_gates = reinterpret_cast<const GlGateDesc *>(c.d_gates.p);
            gp.d_immediates = c.d_imms.p;
            gp.num_gates = c.num_gates, gp.num_selectors = c.num_selectors;
            memcpy(gp.public_inputs_hash, pih, 32);
            a.gate_program = &gp;
        }
        TRY(gl_compute_quotient_polys(&a, quotient.p, ctx));
    }
    TRY(st.mark(3));
    // split into degree-n chunks (prover.rs:153-166) and commit from coefficients
    Batch quot;
    {
        DevBuf chunks;
        if (qdf == (1u << qdb)) {
            chunks = std::move(quotient);  // [nch][n << qdb] read flat is [nch * qdf][n]
        } else {
            TRY(chunks.alloc((uint64_t)nch * qdf * n));
            std::vector<uint64_t> tail((n << qdb) - (uint64_t)qdf * n);
            for (uint32_t k = 0; k < nch; k++) {
                TRY(gl_memcpy_d2h(tail.data(), quotient.p + ((uint64_t)k << (db + qdb)) + (uint64_t)qdf * n, tail.size() * 8, ctx));
                for (uint64_t t : tail)
                    if (t) return fail("Quotient has failed, the vanishing polynomial is not divisible by Z_H");
                TRY(gl_memcpy_d2d(chunks.p + (uint64_t)k * qdf * n, quotient.p + ((uint64_t)k << (db + qdb)), 8ull * qdf * n, ctx));
            }
        }
        TRY(commit(&quot, std::move(chunks), false, nch * qdf, c, ctx, d_salts ? d_salts + 2ull * SALT_SIZE * n_ext : nullptr, false));
    }
    TRY(st.mark(4));
    TRY(step({GlObserveSrc{quot.cap_d.p, cap_words, 0}}, 2, zeta_s));
    TRY(fetch(zeta_s));
    TRY(stream_sync(ctx));
    const E2 zeta{H[zeta_s.off], H[zeta_s.off + 1]};
    if (E2 zn = e2_pow(zeta, n); zn.a == 1 && zn.b == 0) return fail("Opening point is in the subgroup.");
    const uint64_t g = glh::root_of_unity(db);
    const E2 g_zeta = e2_mul(E2{g, 0}, zeta);
    // OpeningSet::new (plonk/proof.rs:305-334): every oracle's polynomials at zeta, the Zs also at g * zeta, left in device memory
    const Batch *oracles[4] = {&c.cs, &wires, &zs, &quot};
    {
        const uint64_t pts[4] = {zeta.a, zeta.b, g_zeta.a, g_zeta.b};
        for (int o = 0; o < 4; o++)
            TRY(gl_eval_polys_ext2(oracles[o]->coeffs.p, oracles[o]->n_polys, db, n, pts, o == 2 ? 2 : 1, D + opens[o].off, ctx));
    }
    TRY(st.mark(5));
    // to_fri_openings (proof.rs:336-356): [constants, sigmas, wires, zs, partial products, quotient], then zs_next; then the FRI alpha
    // ---- PolynomialBatch::prove_openings (fri/oracle.rs:1047-1112) ----
    TRY(step({GlObserveSrc{D + opens[0].off, 2ull * n_polys[0], 0}, GlObserveSrc{D + opens[1].off, 2ull * n_polys[1], 0},
              GlObserveSrc{D + opens[2].off, 2ull * n_polys[2], 0}, GlObserveSrc{D + opens[3].off, 2ull * n_polys[3], 0},
              GlObserveSrc{D + opens[2].off + 2ull * n_polys[2], 2ull * nch, 0}},
             2, alpha_fri_s));
    TRY(fetch(alpha_fri_s));
    TRY(stream_sync(ctx));
    const E2 alpha{H[alpha_fri_s.off], H[alpha_fri_s.off + 1]};
    DevBuf final_poly;  // planar [2][n]
    TRY(final_poly.alloc(2 * n));
    {
        // batch 0: every polynomial of the four oracles at zeta; batch 1: the Zs at g*zeta (circuit_data.rs:351-371)
        std::vector<const uint64_t *> ptrs;
        for (int o = 0; o < 4; o++)
            for (uint32_t k = 0; k < oracles[o]->n_polys; k++) ptrs.push_back(oracles[o]->coeffs.p + (uint64_t)k * n);
        const uint32_t m0 = (uint32_t)ptrs.size();
        for (uint32_t k = 0; k < nch; k++) ptrs.push_back(zs.coeffs.p + (uint64_t)k * n);
        DevBuf d_ptrs, comp;
        TRY(d_ptrs.alloc(ptrs.size()));
        TRY(gl_memcpy_h2d(d_ptrs.p, ptrs.data(), ptrs.size() * 8, ctx));
        TRY(comp.alloc(2 * n));
        const uint64_t al[2] = {alpha.a, alpha.b};
        const struct {
            uint32_t off, m;
            E2 point;
        } batches[2] = {{0, m0, zeta}, {m0, nch, g_zeta}};
        for (int b = 0; b < 2; b++) {
            TRY(gl_fri_reduce_polys_base(reinterpret_cast<const uint64_t *const *>(d_ptrs.p) + batches[b].off, batches[b].m, n, al, comp.p,
                                         ctx));
            const E2 sc = e2_pow(alpha, batches[b].m);  // alpha.shift_poly (util/reducing.rs:103-106)
            const uint64_t pt[2] = {batches[b].point.a, batches[b].point.b}, scale[2] = {sc.a, sc.b};
            TRY(gl_fri_divide_by_linear(comp.p, n, pt, scale, b != 0, final_poly.p, ctx));
        }
        // d_ptrs / comp return to the pool here while their kernels may still be queued: stream order
    }
    TRY(st.mark(6));
    // ---- fri_committed_trees (fri/prover.rs:77-120): no host synchronisation inside — the betas stay on the device ----
    struct Layer {
        DevBuf rows, digests, cap_d;
    };
    std::vector<Layer> layers(n_fri);
    DevBuf final_coeffs_d;
    {
        DevBuf coeffs = std::move(final_poly), vals;
        uint64_t len = n, shift = 7;
        auto lde = [&](DevBuf *dst) -> GlError {
            uint32_t lg = 0;
            while ((1ull << lg) < len) lg++;
            TRY(dst->alloc(2 * (len << c.rate_bits)));
            return gl_coset_lde_batch(coeffs.p, dst->p, 2, lg, c.rate_bits, shift, len, len << c.rate_bits, ctx);
        };
        if (!layers.empty()) TRY(lde(&vals));
        for (uint32_t li = 0; li < n_fri; li++) {
            const uint32_t ab = c.arity_bits[li];
            const uint64_t lde_len = len << c.rate_bits;
            Layer &L = layers[li];
            TRY(L.rows.alloc(2 * lde_len));
            TRY(gl_ext2_interleave(vals.p, lde_len, L.rows.p, ctx));
            TRY(L.digests.alloc(8 * (fs[li].n_leaves - (1ull << c.cap_height)) + 4));
            TRY(L.cap_d.alloc(cap_words));
            TRY(gl_merkle_tree_from_leaves(L.rows.p, fs[li].leaf_len, fs[li].n_leaves, c.cap_height, L.digests.p, L.cap_d.p, ctx));
            TRY(gl_memcpy_d2d(D + fri_caps.off + li * cap_words, L.cap_d.p, cap_words * 8, ctx));
            TRY(step({GlObserveSrc{L.cap_d.p, cap_words, 0}}, 2, Span{fri_betas.off + 2ull * li, 2}));
            DevBuf next;
            TRY(next.alloc(2 * (len >> ab)));
            TRY(gl_fri_fold_device(coeffs.p, len, ab, D + fri_betas.off + 2ull * li, next.p, ctx));
            coeffs = std::move(next);  // the old coefficients return to the pool (stream order keeps them valid)
            len >>= ab;
            shift = glh::pow(shift, 1ull << ab);
            if (li + 1 < n_fri) TRY(lde(&vals));
        }
        // observe_extension_elements(final_poly.coeffs) (fri/prover.rs:117): the two planes read interleaved
        TRY(step({GlObserveSrc{coeffs.p, 2 * len, len}}, 0, Span{}));
        final_coeffs_d = std::move(coeffs);
    }
    TRY(st.mark(7));
    // ---- fri_proof_of_work (fri/prover.rs:122-171) ----
    uint64_t pow_witness = 0;
    TRY(gl_fri_proof_of_work_device(D + T.off, c.pow_bits, D + pow_w.off, &pow_witness, ctx));  // F::order() has 64 bits: leading zeros of the u64 response
    // observe the witness, draw the response, then the query indices (fri/prover.rs:163-170, 181-190)
    TRY(step({GlObserveSrc{D + pow_w.off, 1, 0}}, 1 + nq, resp_idx));
    TRY(st.mark(8));
    // ---- fri_prover_query_rounds (fri/prover.rs:173-260): the indices never leave the device ----
    const uint64_t *d_idx = D + resp_idx.off + 1;
    for (int o = 0; o < 4; o++)  // salted leaves go into the proof whole (fri/prover.rs:203-210)
        TRY(gl_merkle_open_batch_device(oracles[o]->lde.p, 1, n_ext, oracles[o]->leaf_len, n_ext, c.cap_height, oracles[o]->digests.p, d_idx, nq, 0,
                                        D + q_leaves[o].off, D + q_sib[o].off, ctx));
    for (uint32_t li = 0; li < n_fri; li++)
        TRY(gl_merkle_open_batch_device(layers[li].rows.p, fs[li].leaf_len, 1, fs[li].leaf_len, fs[li].n_leaves, c.cap_height, layers[li].digests.p,
                                        d_idx, nq, fs[li].shift, D + s_leaves[li].off, D + s_sib[li].off, ctx));
    // everything the proof consists of, in one go
    TRY(gl_memcpy_d2d(D + caps[0].off, wires.cap_d.p, cap_words * 8, ctx));
    TRY(gl_memcpy_d2d(D + caps[1].off, zs.cap_d.p, cap_words * 8, ctx));
    TRY(gl_memcpy_d2d(D + caps[2].off, quot.cap_d.p, cap_words * 8, ctx));
    TRY(gl_memcpy_d2d(D + final_s.off, final_coeffs_d.p, 2 * final_len * 8, ctx));
    TRY(fetch(Span{fetch0.off, top - fetch0.off}));
    TRY(gl_ctx_synchronize(ctx));
    TRY(st.mark(9));
    if (c.pow_bits && (H[resp_idx.off] >> (64 - c.pow_bits)) != 0) return fail("proof-of-work response does not have the required leading zeros");
    if (H[pow_w.off] != pow_witness) return fail("proof-of-work witness changed between the search and the transcript");
    // ---- write_proof_with_public_inputs (util/serialization.rs:641-689) ----
    Bytes out;
    for (int o = 0; o < 3; o++) out.fields(H + caps[o].off, cap_words);  // wires, zs / partial products, quotient
    // write_opening_set (:557-571): constants, sigmas, wires, zs, zs_next, partial products, quotient
    const uint64_t *ev[4] = {H + opens[0].off, H + opens[1].off, H + opens[2].off, H + opens[3].off};
    out.fields(ev[0], 2ull * n_polys[0]);                                   // constants then sigmas: contiguous
    out.fields(ev[1], 2ull * n_polys[1]);                                   // wires
    out.fields(ev[2], 2ull * nch);                                          // plonk_zs
    out.fields(ev[2] + 2ull * n_polys[2], 2ull * nch);                      // plonk_zs_next
    out.fields(ev[2] + 2ull * nch, 2ull * n_polys[2] - 2ull * nch);         // partial_products
    out.fields(ev[3], 2ull * n_polys[3]);                                   // quotient_polys
    for (uint32_t li = 0; li < n_fri; li++) out.fields(H + fri_caps.off + li * cap_words, cap_words);
    for (uint32_t q = 0; q < nq; q++) {
        for (int o = 0; o < 4; o++) {
            out.fields(H + q_leaves[o].off + (uint64_t)q * leaf_len[o], leaf_len[o]);
            out.merkle_proof(H + q_sib[o].off + (uint64_t)q * init_layers * 4, init_layers);
        }
        for (uint32_t li = 0; li < n_fri; li++) {
            out.fields(H + s_leaves[li].off + (uint64_t)q * fs[li].leaf_len, fs[li].leaf_len);
            out.merkle_proof(H + s_sib[li].off + (uint64_t)q * fs[li].layers * 4, fs[li].layers);
        }
    }
    for (uint64_t i = 0; i < final_len; i++) out.field(H[final_s.off + i]), out.field(H[final_s.off + final_len + i]);  // interleaved (a_i, b_i)
    out.field(pow_witness);
    out.fields(h_public_inputs, num_public_inputs);
    uint8_t *buf = static_cast<uint8_t *>(malloc(out.v.size() ? out.v.size() : 1));
    if (!buf) return fail("out of memory");
    memcpy(buf, out.v.data(), out.v.size());
    *proof = buf;
    *proof_len = out.v.size();
    return st.mark(10);
}

GlError gl_prove(const void *circuit, const uint64_t *d_wires, const uint64_t *h_public_inputs, uint32_t num_public_inputs,
                 uint8_t **proof, uint64_t *proof_len, double *h_stage_ms, void *ctx) {
    return prove_impl(circuit, d_wires, h_public_inputs, num_public_inputs, nullptr, proof, proof_len, h_stage_ms, ctx);
}

// A batch of independent proofs of one circuit with several in flight (configs[4]'s unit of work per GPU): worker w — a host thread
// this call starts — proves witnesses w, w + in_flight, w + 2 in_flight, .. on ctxs[w]. The workers share the circuit handle (a buffer
// pool per context; the launches of its gate kernel take turns) and run at the same time: each fills the other's latency-bound phases.
GlError gl_prove_many(const void *circuit, const uint64_t *const *d_wires, const uint64_t *const *h_public_inputs, uint32_t num_public_inputs,
                      uint32_t count, uint8_t **proofs, uint64_t *proof_lens, void *const *ctxs, uint32_t in_flight) {
    if (!circuit || !proofs || !proof_lens || !ctxs || (count && (!d_wires || (num_public_inputs && !h_public_inputs)))) return fail("null pointer");
    if (in_flight == 0 || in_flight > 16) return fail("in_flight must be 1..16");
    for (uint32_t w = 0; w < in_flight; w++) {
        if (!ctxs[w]) return fail("null context");
        for (uint32_t v = 0; v < w; v++)
            if (ctxs[v] == ctxs[w]) return fail("every worker needs a context of its own");
    }
    for (uint32_t i = 0; i < count; i++) proofs[i] = nullptr, proof_lens[i] = 0;
    std::vector<GlError> errs(in_flight, GlError{0, nullptr});
    auto work = [&](uint32_t w) {
        for (uint32_t i = w; i < count; i += in_flight) {
            GlError e = prove_impl(circuit, d_wires[i], num_public_inputs ? h_public_inputs[i] : nullptr, num_public_inputs, nullptr, &proofs[i], &proof_lens[i],
                                   nullptr, ctxs[w]);
            if (e.code != 0) {
                errs[w] = e;
                return;
            }
        }
    };
    std::vector<std::thread> threads;
    for (uint32_t w = 1; w < in_flight && w < count; w++) threads.emplace_back(work, w);
    work(0);
    for (auto &t : threads) t.join();
    GlError first{0, nullptr};
    for (uint32_t w = 0; w < in_flight; w++)
        if (errs[w].code != 0) {
            if (first.code == 0) first = errs[w];
            else free(errs[w].message);
        }
    if (first.code != 0)
        for (uint32_t i = 0; i < count; i++) {  // all or nothing
            free(proofs[i]);
            proofs[i] = nullptr, proof_lens[i] = 0;
        }
    return first;
}

GlError gl_prove_zk(const void *circuit, const uint64_t *d_wires, const uint64_t *h_public_inputs, uint32_t num_public_inputs,
                    const uint64_t *d_salts, uint8_t **proof, uint64_t *proof_len, double *h_stage_ms, void *ctx) {
    if (!d_salts) return fail("gl_prove_zk: null salt columns");
    return prove_impl(circuit, d_wires, h_public_inputs, num_public_inputs, d_salts, proof, proof_len, h_stage_ms, ctx);
}

}  // extern "C"
