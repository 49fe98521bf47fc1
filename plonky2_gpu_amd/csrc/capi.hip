// capi.hip — the extern "C" boundary of libplonky2_hip.so (declared in include/plonky2_hip.h).
#include <algorithm>
#include <list>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/plonky2_hip.h"
#include "gl_field.h"
#include "knobs.h"
#include "merkle.h"
#include "ntt.h"
#include "ntt_kernels.h"
#include "plonk.h"
#include "fri.h"
#include "ed25519_gate_program.inc"

using namespace plonky2_hip;

namespace {

struct Streams {  // == CudaInnerContext {stream, stream2} (plonky2/src/fri/oracle.rs:43-47)
    hipStream_t stream;
    hipStream_t stream2;
};

GlError ok() { return GlError{0, nullptr}; }

GlError fail(int code, const std::string &msg) { return GlError{code, strdup(msg.c_str())}; }

GlError hip_fail(hipError_t e, const char *what) {
    return fail((int)e, std::string(what) + ": " + hipGetErrorString(e));
}

#define HIP_TRY(expr)                                      \
    do {                                                   \
        hipError_t _e = (expr);                            \
        if (_e != hipSuccess) return hip_fail(_e, #expr);  \
    } while (0)

// Per-device table registry (twiddles are data-independent, a few hundred KiB).
struct CosetEntry {
    CosetTables ct;
    uint64_t last_use = 0;
    uint32_t pins = 0;  // callers between get_coset_tables() and the end of their enqueues
};
struct DeviceState {
    bool have_tables = false;
    NttTables tables;              // twl / twh only: the workspace belongs to a context (CtxState)
    std::list<CosetEntry> cosets;  // LRU cache keyed by (log_n, rate_bits, shift); addresses are stable
    uint64_t coset_tick = 0;
    GateKernel *ed25519_kernel = nullptr;  // the reference symbol compute_quotient_polys' circuit, built on first use
    std::mutex ref_mu;                     // compute_quotient_polys calls on this device take turns (one kernel object, one staging buffer)
    uint64_t *ref_staging = nullptr;       // its column-major staging copy of the three leaf-major inputs
    uint64_t ref_staging_elems = 0;
    bool ref_staging_owned = false;        // false: handed over by gl_reference_quotient_set_staging
};
constexpr size_t COSET_CACHE_ENTRIES = 64;  // a prover uses a handful (one shift, a few sizes); 80 KiB each at 2^18 x 8
std::mutex g_mu;
DeviceState g_dev[64];

// Everything mutable that a call touches besides the caller's buffers belongs to the CONTEXT: the workspace of the natural-order
// multi-pass transforms (also the scans' totals, the openings' partial sums, the transcript's state), the event pair, the
// low-priority hashing stream of the pipelined commit and its events. Two contexts on one device therefore share only read-only
// tables, and their calls run concurrently — two proofs in flight fill each other's latency-bound phases (transcript, small tree
// layers, openings). Keyed by the context's first stream, so that a caller-built {stream, stream2} pair (the reference's
// CudaInnerContext, fri/oracle.rs:43-47) gets its state on first use; gl_ctx_destroy / gl_ctx_release give it back.
struct CtxState {
    int dev = 0;
    NttTables tb;  // twl / twh of the device, scratch of this context
    bool scratch_owned = false;
    hipEvent_t ev[2] = {nullptr, nullptr};
    hipStream_t hash_stream = nullptr;     // pipelined commit: leaf hashing trails the LDE on this lower-priority stream
    std::vector<hipEvent_t> chunk_events;  //   one event per column chunk + one for "tree done"
    bool have_pih = false;                 // gl_reference_set_public_inputs_hash_ctx
    uint64_t pih[4] = {0, 0, 0, 0};
};
std::map<std::pair<int, hipStream_t>, CtxState *> g_ctx;  // g_mu; keyed by (device, first stream): the null stream exists on every device

hipError_t device_tables(int dev, const NttTables **out) {  // g_mu held
    DeviceState &st = g_dev[dev & 63];
    if (!st.have_tables) {
        hipError_t e = ntt_tables_create(&st.tables);
        if (e != hipSuccess) return e;
        st.have_tables = true;
    }
    *out = &st.tables;
    return hipSuccess;
}

void ctx_state_free(CtxState *c) {  // the caller has synchronised the context's streams
    if (c->scratch_owned && c->tb.scratch) (void)hipFree(c->tb.scratch);
    for (hipEvent_t e : c->ev)
        if (e) (void)hipEventDestroy(e);
    if (c->hash_stream) {
        (void)hipStreamSynchronize(c->hash_stream);
        (void)hipStreamDestroy(c->hash_stream);
    }
    for (hipEvent_t e : c->chunk_events) (void)hipEventDestroy(e);
    delete c;
}

Streams *S(void *ctx) { return static_cast<Streams *>(ctx); }

// The state of `ctx` on the current device (DeviceCall has made the context's device current), created on first use.
hipError_t ctx_state(void *ctx, CtxState **out) {
    if (!ctx) return hipErrorInvalidValue;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_ctx.find({dev, S(ctx)->stream});
    if (it == g_ctx.end()) {
        const NttTables *dt;
        e = device_tables(dev, &dt);
        if (e != hipSuccess) return e;
        CtxState *c = new CtxState();
        c->dev = dev;
        c->tb = *dt;
        c->tb.scratch_elems = NTT_SCRATCH_ELEMS;
        e = hipMalloc(&c->tb.scratch, NTT_SCRATCH_ELEMS * sizeof(uint64_t));
        if (e != hipSuccess) {
            delete c;
            return e;
        }
        c->scratch_owned = true;
        it = g_ctx.emplace(std::make_pair(dev, S(ctx)->stream), c).first;
    }
    *out = it->second;
    return hipSuccess;
}

void ctx_state_release(void *ctx) {
    if (!ctx) return;
    CtxState *c = nullptr;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return;  // gl_ctx_release has made the context's device current
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_ctx.find({dev, S(ctx)->stream});
        if (it == g_ctx.end()) return;
        c = it->second;
        g_ctx.erase(it);
    }
    ctx_state_free(c);
}

hipError_t get_tables(void *ctx, const NttTables **out) {
    CtxState *c;
    hipError_t e = ctx_state(ctx, &c);
    if (e != hipSuccess) return e;
    *out = &c->tb;
    return hipSuccess;
}

hipError_t get_events(void *ctx, hipEvent_t *a, hipEvent_t *b) {
    CtxState *c;
    hipError_t e = ctx_state(ctx, &c);
    if (e != hipSuccess) return e;
    if (!c->ev[0]) {  // only the context's own caller thread gets here
        for (int i = 0; i < 2; i++) {
            e = hipEventCreateWithFlags(&c->ev[i], hipEventDisableTiming);
            if (e != hipSuccess) return e;
        }
    }
    *a = c->ev[0];
    *b = c->ev[1];
    return hipSuccess;
}

// Holds one cache entry pinned while its owner enqueues the kernels that read it; an unpinned entry may be
// evicted, and eviction synchronises the device first, so work already enqueued on any stream is safe too.
class CosetLease {
public:
    CosetLease() = default;
    CosetLease(const CosetLease &) = delete;
    CosetLease &operator=(const CosetLease &) = delete;
    ~CosetLease() { release(); }
    const CosetTables &operator*() const { return entry_->ct; }
    void acquire(CosetEntry *e) {  // g_mu held
        release_locked();
        entry_ = e;
        e->pins++;
    }
    void release() {
        if (!entry_) return;
        std::lock_guard<std::mutex> lk(g_mu);
        release_locked();
    }

private:
    void release_locked() {
        if (entry_) entry_->pins--;
        entry_ = nullptr;
    }
    CosetEntry *entry_ = nullptr;
};

hipError_t get_coset_tables(uint32_t log_n, uint32_t rate_bits, uint64_t shift, hipStream_t stream, CosetLease *out) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lk(g_mu);
    DeviceState &st = g_dev[dev & 63];
    for (auto &c : st.cosets)
        if (c.ct.log_n == log_n && c.ct.rate_bits == rate_bits && c.ct.shift == shift) {
            c.last_use = ++st.coset_tick;
            out->acquire(&c);
            return hipSuccess;
        }
    // Miss on a full cache: drop the least recently used entry nobody holds. If every entry is pinned
    // (more concurrent callers than entries) the cache grows instead: a full cache is never an error.
    while (st.cosets.size() >= COSET_CACHE_ENTRIES) {
        auto victim = st.cosets.end();
        for (auto it = st.cosets.begin(); it != st.cosets.end(); ++it)
            if (it->pins == 0 && (victim == st.cosets.end() || it->last_use < victim->last_use)) victim = it;
        if (victim == st.cosets.end()) break;
        e = hipDeviceSynchronize();  // kernels enqueued by earlier, already-returned calls may still read it
        if (e != hipSuccess) return e;
        coset_tables_destroy(&victim->ct);
        st.cosets.erase(victim);
    }
    CosetEntry entry;
    e = coset_tables_create(&entry.ct, log_n, rate_bits, shift, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);  // built on `stream`; other streams may use them later
    if (e != hipSuccess) {
        coset_tables_destroy(&entry.ct);
        return e;
    }
    entry.last_use = ++st.coset_tick;
    st.cosets.push_back(entry);
    out->acquire(&st.cosets.back());
    return hipSuccess;
}

// The gate kernel of the one circuit the reference's compute_quotient_polys is compiled for (ed25519_gate_program.inc).
// A cold hiprtc build of it takes about a minute (gate_jit.hip keeps compiled code objects under
// $PLONKY2_HIP_KERNEL_CACHE; comgr's own cache cuts a repeat to ~2 s), so this has its own lock: table lookups of
// other threads do not wait for it.
std::mutex g_ref_mu;
uint64_t g_ref_pih[4] = {ED25519_REFERENCE_PUBLIC_INPUTS_HASH[0], ED25519_REFERENCE_PUBLIC_INPUTS_HASH[1],
                         ED25519_REFERENCE_PUBLIC_INPUTS_HASH[2], ED25519_REFERENCE_PUBLIC_INPUTS_HASH[3]};

GlError get_ed25519_kernel(const GateKernel **out) {
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(g_ref_mu);
    DeviceState &st = g_dev[dev & 63];
    if (!st.ed25519_kernel) {
        std::string err;
        st.ed25519_kernel = gate_kernel_build(ED25519_INSTRS, ED25519_NUM_INSTRS, ED25519_GATES, ED25519_NUM_GATES, ED25519_IMMEDIATES,
                                              ED25519_NUM_IMMEDIATES, ED25519_NUM_SELECTORS, ED25519_NUM_GATE_CONSTRAINTS,
                                              ED25519_NUM_CHALLENGES, &err);
        if (!st.ed25519_kernel) return fail(GL_E_INVALID, "compute_quotient_polys: building the ed25519 gate kernel failed: " + err);
    }
    *out = st.ed25519_kernel;
    return ok();
}

// Device of a context = device of its first stream; makes it the calling thread's current device.
bool ctx_device(void *ctx, int *dev) {
    if (ctx && S(ctx)->stream) {
        hipDevice_t d;
        if (hipStreamGetDevice(S(ctx)->stream, &d) != hipSuccess) return false;
        int cur = -1;
        if (hipGetDevice(&cur) != hipSuccess) return false;
        if (cur != (int)d && hipSetDevice((int)d) != hipSuccess) return false;
        *dev = (int)d;
        return true;
    }
    return hipGetDevice(dev) == hipSuccess;
}

// Every entry point that takes a ctx runs on the device its context's streams belong to, whatever device the calling thread
// has current: tables, workspace and every allocation made inside the call follow it (a context created on device 1 and used
// from a thread whose current device is still 0 must not touch device 0's state). The device comes from the stream itself, so
// a caller-built {stream, stream2} pair (the reference's CudaInnerContext) works too. Nothing is locked and nothing is ordered
// across contexts (up to round 5 the workspace and the event pair existed once per device and contexts took turns): a context
// is used by one host thread at a time, different contexts by different threads at the same time; data shared between two
// contexts is the caller's to order, as with any two streams.
class DeviceCall {
public:
    explicit DeviceCall(void *ctx) {
        int dev = 0;
        if (!ctx_device(ctx, &dev)) (void)hipGetLastError();
    }
    DeviceCall(const DeviceCall &) = delete;
    DeviceCall &operator=(const DeviceCall &) = delete;
};

__global__ void bit_reverse_columns_kernel(uint64_t *v, uint32_t log_n, uint64_t total) {
    uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= total) return;
    uint64_t n = 1ull << log_n, i = g & (n - 1), base = g - i;
    uint64_t j = log_n ? (__brevll(i) >> (64 - log_n)) : 0;
    if (i < j) {
        uint64_t a = v[base + i], b = v[base + j];
        v[base + i] = b;
        v[base + j] = a;
    }
}


// Element-wise field ops, exported only so that the parity tests can drive gl_field.h with the
// reference's edge operands (field/src/prime_field_testing.rs:7-17).
template <int K>
__device__ uint64_t pow2_case(uint64_t x, int k) {
    if constexpr (K >= 192) {
        return 0;
    } else {
        return k == K ? gl::mul_pow2<K>(x) : pow2_case<K + 1>(x, k);
    }
}

// The deferred-rare-path forms (gl_field.h add_f / sub_f / mul_f / mul_pow2_f) the way the NTT passes use them: a GROUP of three
// independent operations on (x, y), (y, x), (x ^ y, x), one branch, the corrections behind it. `which` picks the member whose result is
// returned, so that the tests see the flagged operation first, in the middle and last in its group, beside unflagged neighbours.
template <int KIND>
__device__ uint64_t deferred_group(uint64_t x, uint64_t y, int which) {
    uint64_t a[3] = {x, y, x ^ y}, b[3] = {y, x, x}, r[3];
    gl::rare_mask f[3];
#pragma unroll
    for (int k = 0; k < 3; k++) r[k] = KIND == 0 ? gl::add_f(a[k], b[k], f[k]) : KIND == 1 ? gl::sub_f(a[k], b[k], f[k]) : gl::mul_f(a[k], b[k], f[k]);
    if (GL_RARE_ANY(f[0] | f[1] | f[2])) {
#pragma unroll
        for (int k = 0; k < 3; k++) r[k] = KIND == 0 ? gl::add_fix(r[k], f[k]) : KIND == 1 ? gl::sub_fix(r[k], f[k]) : gl::mul_fix(r[k], f[k]);
    }
    return which == 0 ? r[0] : which == 1 ? r[1] : r[2];
}
template <int K>
__device__ uint64_t pow2f_case(uint64_t x, int k, bool alone) {
    if constexpr (K >= 96) {
        return 0;
    } else {
        if (k != K) return pow2f_case<K + 1>(x, k, alone);
        gl::rare_mask f0, f1;
        uint64_t r0 = gl::mul_pow2_f<K>(x, f0), r1 = gl::mul_pow2_f<K>(~x, f1);
        if (GL_RARE_ANY(f0 | f1)) r0 = gl::mul_pow2_fix<K>(r0, f0), r1 = gl::mul_pow2_fix<K>(r1, f1);
        return alone ? r0 : gl::add(r0, r1);  // x 2^K + ~x 2^K = (2^64 - 1) 2^K
    }
}

// A lazy-dot-product accumulator built from two test words, so that the parity tests can reach the
// reduction's rare wrap corrections directly (random Poseidon states hit them with probability ~2^-32).
//   mode 0: every field wide (a0 = x, a1 = y, a2 = ~x + (y << 13), small counters from the top bits)
//   mode 1: a0 = x, the other five fields packed into y as small numbers:
//           a1 = y[0:16), a2 = y[16:32), k0 = y[32:40), k1 = y[40:48), k2 = y[48:56)
__device__ gl::DotAcc dotacc_from(int mode, uint64_t x, uint64_t y) {
    gl::DotAcc d;
    if (mode == 0) {
        d.a0 = x, d.a1 = y, d.a2 = ~x + (y << 13);
        d.k0 = (uint32_t)(y >> 59), d.k1 = (uint32_t)(x >> 58), d.k2 = (uint32_t)((x ^ y) & 7);
    } else {
        d.a0 = x, d.a1 = y & 0xFFFF, d.a2 = (y >> 16) & 0xFFFF;
        d.k0 = (uint32_t)(y >> 32) & 0xFF, d.k1 = (uint32_t)(y >> 40) & 0xFF, d.k2 = (uint32_t)(y >> 48) & 0xFF;
    }
    return d;
}

// What a streaming kernel can move on this device (the measured roof next to the 8 TB/s specification): 16 B per lane,
// eight independent pieces per thread in flight, non-temporal loads and stores (the bytes are touched once: default-policy
// accesses reach 4.9-5.6 TB/s in the same shape, non-temporal ones 6.2 TB/s = the guide's 6.3 figure;
// tools/ubench_mem.hip, profiles/r02_ubench_mem.txt).
constexpr int COPY_UNROLL = 8;
__global__ __launch_bounds__(256) void copy16_kernel(const uint64_t *__restrict__ in, uint64_t *__restrict__ out, uint64_t n16) {
    const uint64_t base = (uint64_t)blockIdx.x * (256 * COPY_UNROLL) + threadIdx.x;
    uint64_t v[COPY_UNROLL][2];
#pragma unroll
    for (int u = 0; u < COPY_UNROLL; u++) {
        const uint64_t i = base + (uint64_t)u * 256;
        if (i < n16) {
            v[u][0] = __builtin_nontemporal_load(in + 2 * i);
            v[u][1] = __builtin_nontemporal_load(in + 2 * i + 1);
        }
    }
#pragma unroll
    for (int u = 0; u < COPY_UNROLL; u++) {
        const uint64_t i = base + (uint64_t)u * 256;
        if (i < n16) {
            __builtin_nontemporal_store(v[u][0], out + 2 * i);
            __builtin_nontemporal_store(v[u][1], out + 2 * i + 1);
        }
    }
}

__global__ void field_op_kernel(int op, const uint64_t *a, const uint64_t *b, uint64_t *out, uint64_t n) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t x = a[i], y = b ? b[i] : 0, r = 0;
    switch (op) {
        case 0: r = gl::add(x, y); break;
        case 1: r = gl::sub(x, y); break;
        case 2: r = gl::mul(x, y); break;
        case 3: r = gl::neg(x); break;
        case 4: r = gl::pow7(x); break;
        case 5: r = gl::mac(x, y, y); break;
        case 6: r = pow2_case<0>(x, (int)(y % 192)); break;
        case 7: r = gl::add_canonical(x, gl::canon(y)); break;
        case 8: r = gl::add_c(gl::canon_c(x), gl::canon_c(y)); break;
        case 9: r = gl::sub_c(gl::canon_c(x), gl::canon_c(y)); break;
        case 10: r = gl::mul_c(x, y); break;
        case 11: r = gl::canon_c(x); break;
        case 12: { uint64_t lo, hi; gl::mul_wide(x, y, lo, hi); r = gl::reduce128_c(lo ^ y, hi ^ x) ; } break;
        case 13: r = gl::dot_finish(dotacc_from(0, x, y)); break;
        case 14: r = gl::dot_finish_generic(dotacc_from(0, x, y)); break;
        case 15: r = gl::dot_finish(dotacc_from(1, x, y)); break;
        case 16: r = gl::dot_finish_generic(dotacc_from(1, x, y)); break;
        case 17: r = gl::fold96(x, y & 0x7FFFFFFFFFFFFFFFull); break;  // x + (y mod 2^63) * 2^32: the ACC accumulators' fold
        case 18: case 19: case 20: r = deferred_group<0>(x, y, op - 18); break;
        case 21: case 22: case 23: r = deferred_group<1>(x, y, op - 21); break;
        case 24: case 25: case 26: r = deferred_group<2>(x, y, op - 24); break;
        case 27: r = pow2f_case<0>(x, (int)((uint32_t)y % 96), (y >> 32) != 0); break;
        default: r = x; break;
    }
    out[i] = (op >= 8 && op <= 12) ? r : gl::canon(r);  // canonical-domain ops must already be canonical
}

// out[(q * n_cols + c) * L + i] = lde[c * col_stride + q * L + i]: every column's leaf range of every rank, grouped by
// rank — what a column-sharded commit sends (dist.py); 16 B per lane, L is a multiple of 2.
__global__ __launch_bounds__(256) void pack_leaf_ranges_kernel(const uint64_t *__restrict__ lde, uint64_t col_stride, uint32_t n_cols,
                                                               uint64_t L, uint32_t world, uint64_t *__restrict__ out) {
    const uint64_t pairs = L / 2, total = (uint64_t)world * n_cols * pairs;
    for (uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t i = (g % pairs) * 2, rest = g / pairs;
        const uint32_t c = (uint32_t)(rest % n_cols), q = (uint32_t)(rest / n_cols);
        const uint4 v = *reinterpret_cast<const uint4 *>(lde + (uint64_t)c * col_stride + (uint64_t)q * L + i);
        *reinterpret_cast<uint4 *>(out + ((uint64_t)q * n_cols + c) * L + i) = v;
    }
}

// Pipelined commit (large commitments): the columns are extended chunk by chunk on the caller's stream while a second,
// lower-priority stream absorbs the finished chunks into the leaves' sponges (hash_leaves_chunk). The LDE passes are
// latency-bound and leave half of the vector ALU idle; the hashing is ALU-bound: running them side by
// side hides most of the LDE. PLONKY2_COMMIT_PIPELINE=0 turns it off (A/B measurements).
// The leaf-major copy of a commit is written by the leaf-hashing lanes themselves (merkle.h); PLONKY2_FUSED_LEAVES=0 (diagnostic
// build) goes back to the separate transposition on stream2.
bool fused_leaves_enabled() {
    static const bool v = [] {
        const char *e = PLONKY2_KNOB("PLONKY2_FUSED_LEAVES");
        return !(e && e[0] == '0');
    }();
    return v;
}

bool commit_pipeline_enabled() {
    static const bool v = [] {
        const char *e = PLONKY2_KNOB("PLONKY2_COMMIT_PIPELINE");
        return !(e && e[0] == '0');
    }();
    return v;
}

hipError_t get_hash_stream(void *ctx, hipStream_t *hs, std::vector<hipEvent_t> **events, size_t need) {
    CtxState *c;
    hipError_t e = ctx_state(ctx, &c);
    if (e != hipSuccess) return e;
    if (!c->hash_stream) {
        int lo = 0, hi = 0;  // numerically lower = higher priority; the hashing takes the LOWEST so that the LDE runs ahead
        e = hipDeviceGetStreamPriorityRange(&lo, &hi);
        if (e != hipSuccess) return e;
        e = hipStreamCreateWithPriority(&c->hash_stream, hipStreamNonBlocking, lo);
        if (e != hipSuccess) return e;
    }
    while (c->chunk_events.size() < need) {
        hipEvent_t ev;
        e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
        if (e != hipSuccess) return e;
        c->chunk_events.push_back(ev);
    }
    *hs = c->hash_stream;
    *events = &c->chunk_events;
    return hipSuccess;
}

GlError commit_from_coeffs_impl(const uint64_t *d_coeffs, uint64_t poly_num, uint32_t log_n, uint32_t rate_bits,
                                uint32_t cap_height, uint32_t salt_size, uint64_t shift, uint64_t *d_lde,
                                uint64_t *d_leaves, uint64_t *d_digests, uint64_t *d_cap, Streams *s,
                                bool sync_stream2_before_leaves) {
    if (!d_coeffs || !d_lde || !d_digests || !d_cap || !s) return fail(GL_E_INVALID, "null pointer");
    if (log_n + rate_bits > 32 || cap_height > log_n + rate_bits)
        return fail(GL_E_INVALID, "cap_height should be at most log2(leaves.len())");
    if (poly_num + salt_size == 0 || poly_num + salt_size > 0xFFFFFFFFull) return fail(GL_E_INVALID, "bad poly_num");
    const uint64_t n = 1ull << log_n, n_ext = n << rate_bits;
    const NttTables *tb;
    CosetLease ct;
    HIP_TRY(get_tables(s, &tb));
    HIP_TRY(get_coset_tables(log_n, rate_bits, shift, s->stream, &ct));
    (void)sync_stream2_before_leaves;
    const uint32_t leaf_len = (uint32_t)(poly_num + salt_size);
    uint64_t CHUNK = 16;  // columns per pipeline step: two rate blocks
    if (const char *e = PLONKY2_KNOB("PLONKY2_COMMIT_CHUNK")) {  // diagnostic build: another multiple of 8
        const unsigned long v = strtoul(e, nullptr, 10);
        if (v >= 8 && v <= 1024 && v % 8 == 0) CHUNK = v;
    }
    if (commit_pipeline_enabled() && poly_num >= 3 * CHUNK && n_ext >= (1ull << 16)) {
        const size_t n_chunks = (size_t)((poly_num + CHUNK - 1) / CHUNK);
        hipStream_t hs = nullptr;
        // An error after work has been queued on the hash stream or on stream2 must not leave those kernels running behind
        // the caller's back (its gl_ctx_synchronize and frees only cover its own stream): the failing path waits for both.
        auto pipelined = [&]() -> GlError {
        std::vector<hipEvent_t> *evs;
        HIP_TRY(get_hash_stream(s, &hs, &evs, n_chunks + 2));
        // the hash stream starts behind whatever the caller has queued (the buffers may still be in use by earlier work)
        HIP_TRY(hipEventRecord((*evs)[n_chunks], s->stream));
        HIP_TRY(hipStreamWaitEvent(hs, (*evs)[n_chunks], 0));
        const bool fused = d_leaves && fused_leaves_enabled();
        // The reference's caller passes ONE region as coefficients and as leaves (merkle_tree_from_coeffs(values_device,
        // values_device, ..), fri/oracle.rs:409-422): the coefficients [poly_num][n] occupy the slots of the first
        // ceil(poly_num*n / leaf_len) leaf rows, and the LDE of chunk c+1.. still reads them while chunk c is hashed. The
        // hashing lanes therefore leave those rows alone; one transposition of just these rows (1/2^rate_bits of the copy)
        // runs on the hash stream after the last chunk, i.e. behind the last LDE launch.
        uint64_t rows_from = 0;
        if (fused) {
            const uintptr_t c_lo = (uintptr_t)d_coeffs, c_hi = (uintptr_t)(d_coeffs + poly_num * n);
            const uintptr_t l_lo = (uintptr_t)d_leaves, l_hi = (uintptr_t)(d_leaves + (uint64_t)leaf_len * n_ext);
            if (c_lo < l_hi && l_lo < c_hi) {
                const uint64_t past = (uint64_t)(c_hi - l_lo) / 8;  // u64 slots of the leaf region up to the end of the coefficients
                rows_from = std::min<uint64_t>(n_ext, (past + leaf_len - 1) / leaf_len);
            }
        }
        if (fused) {
            // d_leaves may still be read by what the caller queued on stream2 (the reference's caller has its D2H of the
            // coefficients there, oracle.rs:403-407, and region A is overwritten by the leaves, plonky2_gpu.cu:586)
            hipEvent_t ev_a = nullptr, ev_b = nullptr;
            HIP_TRY(get_events(s, &ev_a, &ev_b));
            HIP_TRY(hipEventRecord(ev_a, s->stream2));
            if (!PLONKY2_KNOB("PLONKY2_DROP_STREAM2_WAIT"))  // diagnostic build: shows that tests/test_gpu_stream2.py notices the loss
                HIP_TRY(hipStreamWaitEvent(hs, ev_a, 0));
        }
        // A launch that starts in the middle of the leaf (c0 != 0) carries only the capacity, so its first block must be a
        // full one: if the last chunk (with the salt columns and a trailing partial block) would be shorter than a rate
        // block, the chunk before it is not absorbed on its own but together with the last.
        const uint64_t last_c0 = (n_chunks - 1) * CHUNK;
        const bool merge_last_two = (leaf_len & 7) && leaf_len - last_c0 < 8;
        uint64_t absorbed = 0;
        for (size_t c = 0; c < n_chunks; c++) {
            const bool last = c + 1 == n_chunks;
            const uint64_t c0 = c * CHUNK, c1 = last ? poly_num : c0 + CHUNK;
            HIP_TRY(coset_lde_batch(*tb, *ct, d_coeffs + c0 * n, d_lde + c0 * n_ext, c1 - c0, n, n_ext, s->stream));
            HIP_TRY(hipEventRecord((*evs)[c], s->stream));
            HIP_TRY(hipStreamWaitEvent(hs, (*evs)[c], 0));
            if (!last && merge_last_two && c + 2 == n_chunks) continue;
            const uint64_t upto = last ? leaf_len : c1;  // the last launch also takes the salt columns (already in d_lde)
            HIP_TRY(hash_leaves_chunk(d_lde, (uint32_t)absorbed, (uint32_t)upto, leaf_len, n_ext, n_ext, cap_height, d_digests, d_cap, hs,
                                      fused ? d_leaves : nullptr, rows_from));
            absorbed = upto;
        }
        if (rows_from) HIP_TRY(transpose_to_leaf_major(d_lde, d_leaves, leaf_len, rows_from, n_ext, hs));
        hipEvent_t ev_lde2 = nullptr, ev_tr2 = nullptr;
        if (d_leaves && !fused) {
            HIP_TRY(get_events(s, &ev_lde2, &ev_tr2));
            HIP_TRY(hipEventRecord(ev_lde2, s->stream));
            HIP_TRY(hipStreamWaitEvent(s->stream2, ev_lde2, 0));
            HIP_TRY(transpose_to_leaf_major(d_lde, d_leaves, leaf_len, n_ext, n_ext, s->stream2));
            HIP_TRY(hipEventRecord(ev_tr2, s->stream2));
        }
        HIP_TRY(merkle_tree_layers(d_digests, d_cap, n_ext, cap_height, hs));
        HIP_TRY(hipEventRecord((*evs)[n_chunks + 1], hs));
        HIP_TRY(hipStreamWaitEvent(s->stream, (*evs)[n_chunks + 1], 0));  // the caller's stream continues after the tree
        if (d_leaves && !fused) HIP_TRY(hipStreamWaitEvent(s->stream, ev_tr2, 0));
        return ok();
        };
        GlError r = pipelined();
        if (r.code != 0) {
            if (hs) (void)hipStreamSynchronize(hs);
            (void)hipStreamSynchronize(s->stream2);
        }
        return r;
    }
    HIP_TRY(coset_lde_batch(*tb, *ct, d_coeffs, d_lde, poly_num, n, n_ext, s->stream));
    hipEvent_t ev_lde = nullptr, ev_tr = nullptr;
    const bool fused = d_leaves && fused_leaves_enabled();
    if (fused) {
        HIP_TRY(get_events(s, &ev_lde, &ev_tr));  // stream2's earlier work (see above) before d_leaves is written
        HIP_TRY(hipEventRecord(ev_lde, s->stream2));
        if (!PLONKY2_KNOB("PLONKY2_DROP_STREAM2_WAIT")) HIP_TRY(hipStreamWaitEvent(s->stream, ev_lde, 0));
    } else if (d_leaves) {
        // The leaf-major copy is pure HBM traffic and the Poseidon hashing pure integer ALU work:
        // run the transpose on stream2, concurrently with the tree on stream. It starts after the LDE
        // (event) and after whatever the caller queued on stream2 before this call — the reference's
        // caller has its D2H of the coefficients there (oracle.rs:403-407), which is exactly what
        // must finish before region A is overwritten (plonky2_gpu.cu:586), now by stream order
        // instead of a host-side stream synchronise.
        HIP_TRY(get_events(s, &ev_lde, &ev_tr));
        HIP_TRY(hipEventRecord(ev_lde, s->stream));
        HIP_TRY(hipStreamWaitEvent(s->stream2, ev_lde, 0));
        HIP_TRY(transpose_to_leaf_major(d_lde, d_leaves, (uint32_t)(poly_num + salt_size), n_ext, n_ext, s->stream2));
        HIP_TRY(hipEventRecord(ev_tr, s->stream2));
    }
    HIP_TRY(merkle_tree_from_columns(d_lde, (uint32_t)(poly_num + salt_size), n_ext, n_ext, cap_height, d_digests, d_cap,
                                     s->stream, fused ? d_leaves : nullptr));
    if (d_leaves && !fused) HIP_TRY(hipStreamWaitEvent(s->stream, ev_tr, 0));
    return ok();
}

}  // namespace

extern "C" {

const char *gl_version(void) { return "plonky2_hip 0.6.0 gfx950"; }

int gl_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

void *gl_ctx_create(int device) {
    if (hipSetDevice(device) != hipSuccess) return nullptr;
    Streams *s = (Streams *)calloc(1, sizeof(Streams));
    if (!s) return nullptr;
    if (hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&s->stream2, hipStreamNonBlocking) != hipSuccess) {
        free(s);
        return nullptr;
    }
    CtxState *c;
    if (ctx_state(s, &c) != hipSuccess) {  // the device's tables and this context's workspace, now rather than inside the first call
        (void)hipGetLastError();
        (void)hipStreamDestroy(s->stream);
        (void)hipStreamDestroy(s->stream2);
        free(s);
        return nullptr;
    }
    return s;
}

void gl_ctx_release(void *ctx) {
    if (!ctx) return;
    int dev = 0;
    if (!ctx_device(ctx, &dev)) (void)hipGetLastError();
    (void)hipStreamSynchronize(S(ctx)->stream);
    (void)hipStreamSynchronize(S(ctx)->stream2);
    ctx_state_release(ctx);
}

void gl_ctx_destroy(void *ctx) {
    if (!ctx) return;
    gl_ctx_release(ctx);
    (void)hipStreamDestroy(S(ctx)->stream);
    (void)hipStreamDestroy(S(ctx)->stream2);
    free(ctx);
}

uint64_t gl_workspace_bytes(void) { return NTT_SCRATCH_ELEMS * sizeof(uint64_t); }

GlError gl_ctx_set_workspace(void *ctx, void *d_workspace, uint64_t bytes) {
    DeviceCall device_call(ctx);
    if (!ctx) return fail(GL_E_INVALID, "null ctx");
    if (d_workspace && (bytes < gl_workspace_bytes() || ((uintptr_t)d_workspace & 15)))
        return fail(GL_E_INVALID, "a caller-provided workspace holds at least gl_workspace_bytes() bytes, 16-byte aligned");
    CtxState *c;
    HIP_TRY(ctx_state(ctx, &c));
    HIP_TRY(hipStreamSynchronize(S(ctx)->stream));  // nothing of this context may still use the old one
    if (c->hash_stream) HIP_TRY(hipStreamSynchronize(c->hash_stream));
    if (c->scratch_owned && c->tb.scratch) HIP_TRY(hipFree(c->tb.scratch));
    c->tb.scratch = nullptr;
    c->scratch_owned = false;
    if (d_workspace) {
        c->tb.scratch = static_cast<uint64_t *>(d_workspace);
        c->tb.scratch_elems = bytes / 8;
    } else {
        HIP_TRY(hipMalloc(&c->tb.scratch, NTT_SCRATCH_ELEMS * sizeof(uint64_t)));
        c->tb.scratch_elems = NTT_SCRATCH_ELEMS;
        c->scratch_owned = true;
    }
    return ok();
}

GlError gl_ctx_synchronize(void *ctx) {
    if (!ctx) return fail(GL_E_INVALID, "null ctx");
    HIP_TRY(hipStreamSynchronize(S(ctx)->stream));
    HIP_TRY(hipStreamSynchronize(S(ctx)->stream2));
    return ok();
}

GlError gl_malloc(void **d_ptr, uint64_t bytes) {
    if (!d_ptr) return fail(GL_E_INVALID, "null pointer");
    HIP_TRY(hipMalloc(d_ptr, bytes ? bytes : 8));
    return ok();
}

GlError gl_ctx_malloc(void **d_ptr, uint64_t bytes, void *ctx) {
    if (!d_ptr || !ctx) return fail(GL_E_INVALID, "null pointer");
    int dev = 0;
    if (!ctx_device(ctx, &dev)) return fail(GL_E_INVALID, "the context's stream has no device");
    HIP_TRY(hipMalloc(d_ptr, bytes ? bytes : 8));
    return ok();
}

GlError gl_free(void *d_ptr) {
    HIP_TRY(hipFree(d_ptr));
    return ok();
}

GlError gl_malloc_host(void **h_ptr, uint64_t bytes) {
    if (!h_ptr) return fail(GL_E_INVALID, "null pointer");
    HIP_TRY(hipHostMalloc(h_ptr, bytes ? bytes : 8, hipHostMallocDefault));
    return ok();
}

GlError gl_free_host(void *h_ptr) {
    HIP_TRY(hipHostFree(h_ptr));
    return ok();
}

GlError gl_memcpy_h2d(void *d_dst, const void *h_src, uint64_t bytes, void *ctx) {
    DeviceCall device_call(ctx);
    if (!ctx) return fail(GL_E_INVALID, "null ctx");
    HIP_TRY(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, S(ctx)->stream));
    HIP_TRY(hipStreamSynchronize(S(ctx)->stream));
    return ok();
}

GlError gl_memcpy_h2d_async(void *d_dst, const void *h_src, uint64_t bytes, void *ctx) {
    DeviceCall device_call(ctx);
    if (!ctx) return fail(GL_E_INVALID, "null ctx");
    HIP_TRY(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, S(ctx)->stream2));
    return ok();
}

GlError gl_memcpy_d2h(void *h_dst, const void *d_src, uint64_t bytes, void *ctx) {
    DeviceCall device_call(ctx);
    if (!ctx) return fail(GL_E_INVALID, "null ctx");
    HIP_TRY(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, S(ctx)->stream));
    HIP_TRY(hipStreamSynchronize(S(ctx)->stream));
    return ok();
}

GlError gl_memcpy_d2d(void *d_dst, const void *d_src, uint64_t bytes, void *ctx) {
    DeviceCall device_call(ctx);
    if (!ctx) return fail(GL_E_INVALID, "null ctx");
    HIP_TRY(hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, S(ctx)->stream));
    return ok();
}

GlError gl_memset_zero(void *d_dst, uint64_t bytes, void *ctx) {
    DeviceCall device_call(ctx);
    if (!ctx) return fail(GL_E_INVALID, "null ctx");
    HIP_TRY(hipMemsetAsync(d_dst, 0, bytes, S(ctx)->stream));
    return ok();
}

GlError gl_event_create(void **event) {
    if (!event) return fail(GL_E_INVALID, "null pointer");
    hipEvent_t ev;
    HIP_TRY(hipEventCreate(&ev));
    *event = ev;
    return ok();
}

GlError gl_event_record(void *event, void *ctx) {
    if (!event || !ctx) return fail(GL_E_INVALID, "null pointer");
    HIP_TRY(hipEventRecord((hipEvent_t)event, S(ctx)->stream));
    return ok();
}

GlError gl_event_elapsed_ms(float *ms, void *start_event, void *stop_event) {
    if (!ms || !start_event || !stop_event) return fail(GL_E_INVALID, "null pointer");
    HIP_TRY(hipEventSynchronize((hipEvent_t)stop_event));
    HIP_TRY(hipEventElapsedTime(ms, (hipEvent_t)start_event, (hipEvent_t)stop_event));
    return ok();
}

void gl_event_destroy(void *event) {
    if (event) (void)hipEventDestroy((hipEvent_t)event);
}

GlError gl_ntt_batch(uint64_t *d_values, uint64_t poly_num, uint32_t log_n, uint64_t stride, int inverse,
                     int bit_reversed, void *ctx) {
    DeviceCall device_call(ctx);
    if (!ctx || (!d_values && poly_num)) return fail(GL_E_INVALID, "null pointer");
    if (log_n > 24) return fail(GL_E_INVALID, "log_n > 24 is not supported by this build");
    if (stride < (1ull << log_n)) return fail(GL_E_INVALID, "stride smaller than the polynomial");
    if (inverse && bit_reversed) return fail(GL_E_INVALID, "bit-reversed inverse is not on the hot path");
    if (inverse && (stride & ((1ull << log_n) - 1))) return fail(GL_E_INVALID, "inverse needs stride % n == 0");
    if ((stride & 1) && log_n > 0 && poly_num > 1) return fail(GL_E_INVALID, "stride must be even (16-byte accesses)");
    if ((uintptr_t)d_values & 15) return fail(GL_E_INVALID, "d_values must be 16-byte aligned");
    const NttTables *tb;
    HIP_TRY(get_tables(ctx, &tb));
    HIP_TRY(ntt_batch(*tb, d_values, d_values, poly_num, log_n, stride, stride,
                      bit_reversed ? NttOrder::BitReversed : NttOrder::Natural, inverse != 0, S(ctx)->stream));
    return ok();
}

GlError gl_coset_lde_batch(const uint64_t *d_coeffs, uint64_t *d_out, uint64_t poly_num, uint32_t log_n,
                           uint32_t rate_bits, uint64_t shift, uint64_t src_stride, uint64_t dst_stride, void *ctx) {
    DeviceCall device_call(ctx);
    if (!ctx || ((!d_coeffs || !d_out) && poly_num)) return fail(GL_E_INVALID, "null pointer");
    if (log_n > 24 || rate_bits > 8) return fail(GL_E_INVALID, "log_n > 24 or rate_bits > 8 not supported");
    const uint64_t n = 1ull << log_n;
    if (src_stride < n || dst_stride < (n << rate_bits)) return fail(GL_E_INVALID, "stride too small");
    if (((uintptr_t)d_coeffs | (uintptr_t)d_out) & 15) return fail(GL_E_INVALID, "buffers must be 16-byte aligned");
    if (log_n > 0 && ((src_stride | dst_stride) & 1)) return fail(GL_E_INVALID, "strides must be even");
    const NttTables *tb;
    CosetLease ct;
    HIP_TRY(get_tables(ctx, &tb));
    HIP_TRY(get_coset_tables(log_n, rate_bits, shift, S(ctx)->stream, &ct));
    HIP_TRY(coset_lde_batch(*tb, *ct, d_coeffs, d_out, poly_num, src_stride, dst_stride, S(ctx)->stream));
    return ok();
}

GlError gl_coset_ntt_batch(uint64_t *d_values, uint64_t poly_num, uint32_t log_n, uint64_t stride, uint64_t shift, int inverse,
                           void *ctx) {
    DeviceCall device_call(ctx);
    if (!ctx || (!d_values && poly_num)) return fail(GL_E_INVALID, "null pointer");
    if (shift % glh::P == 0) return fail(GL_E_INVALID, "shift must be non-zero");
    CosetLease ct;
    if (!inverse) {
        // coset_fft: c_i *= shift^i, then fft (polynomial/mod.rs:286-299)
        HIP_TRY(get_coset_tables(log_n, 0, shift % glh::P, S(ctx)->stream, &ct));
        HIP_TRY(scale_by_powers(*ct, d_values, poly_num, stride, S(ctx)->stream));
        return gl_ntt_batch(d_values, poly_num, log_n, stride, 0, 0, ctx);
    }
    // coset_ifft: ifft, then c_i *= shift^-i (polynomial/mod.rs:64-77)
    GlError e = gl_ntt_batch(d_values, poly_num, log_n, stride, 1, 0, ctx);
    if (e.code) return e;
    HIP_TRY(get_coset_tables(log_n, 0, glh::inv(shift), S(ctx)->stream, &ct));
    HIP_TRY(scale_by_powers(*ct, d_values, poly_num, stride, S(ctx)->stream));
    return ok();
}

GlError gl_permutation_partial_products(const uint64_t *d_wires, uint64_t wires_stride, const uint64_t *d_sigmas,
                                        uint64_t sigmas_stride, const uint64_t *d_k_is, const uint64_t *h_betas,
                                        const uint64_t *h_gammas, uint32_t num_challenges, uint32_t num_routed,
                                        uint32_t quotient_degree_factor, uint32_t log_n, uint64_t *d_out, void *ctx) {
    DeviceCall device_call(ctx);
    if (!ctx || !d_wires || !d_sigmas || !d_k_is || !h_betas || !h_gammas || !d_out) return fail(GL_E_INVALID, "null pointer");
    if (num_challenges == 0 || num_challenges > 4) return fail(GL_E_INVALID, "num_challenges must be 1..4");
    if (quotient_degree_factor < 2 || num_routed == 0) return fail(GL_E_INVALID, "bad num_routed / quotient_degree_factor");
    if (quotient_degree_factor >= num_routed)
        return fail(GL_E_INVALID, "quotient_degree_factor must be smaller than num_routed_wires (prover.rs:102-105)");
    if (log_n > 24) return fail(GL_E_INVALID, "log_n > 24");
    const NttTables *tb;
    HIP_TRY(get_tables(ctx, &tb));
    HIP_TRY(permutation_partial_products(*tb, d_wires, wires_stride, d_sigmas, sigmas_stride, d_k_is, h_betas, h_gammas,
                                         num_challenges, num_routed, quotient_degree_factor, log_n, d_out, S(ctx)->stream));
    return ok();
}

GlError gl_gate_kernel_build(const GlGateInstr *h_instrs, uint32_t num_instrs, const GlGateDesc *h_gates, uint32_t num_gates,
                             const uint64_t *h_immediates, uint32_t num_immediates, uint32_t num_selectors,
                             uint32_t num_gate_constraints, uint32_t num_challenges, void **kernel) {
    if (!h_instrs || !h_gates || !kernel || (num_immediates && !h_immediates)) return fail(GL_E_INVALID, "null pointer");
    if (num_gate_constraints > 256) return fail(GL_E_INVALID, "num_gate_constraints > 256");
    std::string err;
    GateKernel *k = gate_kernel_build(reinterpret_cast<const uint16_t *>(h_instrs), num_instrs, reinterpret_cast<const uint32_t *>(h_gates),
                                      num_gates, h_immediates, num_immediates, num_selectors, num_gate_constraints, num_challenges, &err);
    if (!k) return fail(GL_E_INVALID, err.c_str());
    *kernel = k;
    return ok();
}

void gl_gate_kernel_destroy(void *kernel) { gate_kernel_destroy(static_cast<GateKernel *>(kernel)); }

const char *gl_gate_kernel_source(const void *kernel) { return kernel ? gate_kernel_source(static_cast<const GateKernel *>(kernel)) : ""; }

GlError gl_compute_quotient_polys(const GlQuotientArgs *args, uint64_t *d_quotient_polys, void *ctx) {
    DeviceCall device_call(ctx);
    if (!ctx || !args || !d_quotient_polys) return fail(GL_E_INVALID, "null pointer");
    if (!args->d_wires_leaves || !args->d_constants_sigmas_leaves || !args->d_zs_partial_products_leaves || !args->d_k_is ||
        !args->h_betas || !args->h_gammas || !args->h_alphas)
        return fail(GL_E_INVALID, "null pointer in GlQuotientArgs");
    QuotientArgs a = {};
    a.wires_leaves = args->d_wires_leaves;
    a.cs_leaves = args->d_constants_sigmas_leaves;
    a.zpp_leaves = args->d_zs_partial_products_leaves;
    a.wires_len = args->wires_leaf_len;
    a.cs_len = args->constants_sigmas_leaf_len;
    a.zpp_len = args->zs_partial_products_leaf_len;
    a.k_is = args->d_k_is;
    a.gate_terms = args->d_gate_constraint_terms;
    a.betas = args->h_betas;
    a.gammas = args->h_gammas;
    a.alphas = args->h_alphas;
    a.num_constants = args->num_constants;
    a.num_routed = args->num_routed_wires;
    a.num_challenges = args->num_challenges;
    a.num_gate_constraints = args->num_gate_constraints;
    a.degree_bits = args->degree_bits;
    a.rate_bits = args->rate_bits;
    a.quotient_degree_factor = args->quotient_degree_factor;
    a.shift = args->coset_shift;
    a.column_stride = args->column_stride;
    if (args->gate_kernel) {
        if (args->gate_program || args->d_gate_constraint_terms)
            return fail(GL_E_INVALID, "give one source of gate constraints: terms, a gate program or a gate kernel");
        if (!args->h_public_inputs_hash || !args->d_gate_workspace) return fail(GL_E_INVALID, "gate_kernel needs h_public_inputs_hash and d_gate_workspace");
        a.gate_kernel = static_cast<const GateKernel *>(args->gate_kernel);
        if (gate_kernel_num_constraints(a.gate_kernel) != args->num_gate_constraints || gate_kernel_num_challenges(a.gate_kernel) != args->num_challenges)
            return fail(GL_E_INVALID, "gate kernel was built for other num_gate_constraints / num_challenges");
        a.public_inputs_hash = args->h_public_inputs_hash;
        a.gate_partial_workspace = args->d_gate_workspace;
    }
    GateProgramArgs gpa = {};
    if (args->gate_program) {
        const GlGateProgram *g = args->gate_program;
        if (!g->d_instrs || !g->d_gates) return fail(GL_E_INVALID, "null pointer in GlGateProgram");
        if (args->d_gate_constraint_terms) return fail(GL_E_INVALID, "give either gate terms or a gate program, not both");
        if (args->num_gate_constraints > 256) return fail(GL_E_INVALID, "num_gate_constraints > 256");
        gpa.instrs = reinterpret_cast<const uint16_t *>(g->d_instrs);
        gpa.gates = reinterpret_cast<const uint32_t *>(g->d_gates);
        gpa.imms = g->d_immediates;
        gpa.num_gates = g->num_gates;
        gpa.num_selectors = g->num_selectors;
        for (int k = 0; k < 4; k++) gpa.public_inputs_hash[k] = g->public_inputs_hash[k];
        a.gate_program = &gpa;
    }
    uint32_t qdb = 0;
    while ((1u << qdb) < a.quotient_degree_factor) qdb++;
    if (a.quotient_degree_factor < 2 || qdb > a.rate_bits)
        return fail(GL_E_INVALID, "constraints of degree higher than the rate are not supported (prover.rs:807-811)");
    const NttTables *tb;
    HIP_TRY(get_tables(ctx, &tb));
    hipError_t e = quotient_values(*tb, a, d_quotient_polys, S(ctx)->stream);
    if (e == hipErrorInvalidValue) return fail(GL_E_INVALID, "inconsistent GlQuotientArgs (leaf lengths / counts / sizes)");
    HIP_TRY(e);
    // values on the coset -> coefficients: coset_ifft per challenge (prover.rs:1009-1021)
    const uint32_t log_lde = a.degree_bits + qdb;
    return gl_coset_ntt_batch(d_quotient_polys, a.num_challenges, log_lde, 1ull << log_lde, a.shift, 1, ctx);
}

GlError gl_eval_polys_ext2(const uint64_t *d_coeffs, uint64_t poly_num, uint32_t log_n, uint64_t stride, const uint64_t *h_points,
                           uint32_t num_points, uint64_t *d_out, void *ctx) {
    DeviceCall device_call(ctx);
    if (!ctx || !h_points || !d_out || (!d_coeffs && poly_num)) return fail(GL_E_INVALID, "null pointer");
    if (num_points == 0 || num_points > 4) return fail(GL_E_INVALID, "num_points must be 1..4");
    if (poly_num > 65535) return fail(GL_E_INVALID, "poly_num > 65535");
    if (stride < (1ull << log_n)) return fail(GL_E_INVALID, "stride smaller than the polynomial");
    const NttTables *tb;
    HIP_TRY(get_tables(ctx, &tb));
    hipError_t e = eval_polys_ext2(*tb, d_coeffs, poly_num, log_n, stride, h_points, num_points, d_out, S(ctx)->stream);
    if (e == hipErrorInvalidValue) return fail(GL_E_INVALID, "unsupported size for gl_eval_polys_ext2");
    HIP_TRY(e);
    return ok();
}

GlError gl_fri_reduce_polys_base(const uint64_t *const *d_poly_ptrs, uint32_t num_polys, uint64_t n, const uint64_t *h_alpha,
                                 uint64_t *d_out, void *ctx) {
    DeviceCall device_call(ctx);
    if (!ctx || !d_poly_ptrs || !h_alpha || !d_out) return fail(GL_E_INVALID, "null pointer");
    if (num_polys == 0 || num_polys > (1u << 20) || n == 0) return fail(GL_E_INVALID, "bad sizes");
    const NttTables *tb;
    HIP_TRY(get_tables(ctx, &tb));
    HIP_TRY(fri_reduce_polys_base(*tb, d_poly_ptrs, num_polys, n, h_alpha, d_out, S(ctx)->stream));
    return ok();
}

GlError gl_fri_divide_by_linear(uint64_t *d_composition, uint64_t n, const uint64_t *h_point, const uint64_t *h_scale, int accumulate,
                                uint64_t *d_final, void *ctx) {
    DeviceCall device_call(ctx);
    if (!ctx || !d_composition || !h_point || !h_scale || !d_final) return fail(GL_E_INVALID, "null pointer");
    if (n < 2 || n > (1ull << 30)) return fail(GL_E_INVALID, "bad length");
    const NttTables *tb;
    HIP_TRY(get_tables(ctx, &tb));
    hipError_t e = fri_divide_by_linear_accumulate(*tb, d_composition, n, h_point, h_scale, accumulate, d_final, S(ctx)->stream);
    if (e == hipErrorInvalidValue) return fail(GL_E_INVALID, "point has no inverse / workspace too small");
    HIP_TRY(e);
    return ok();
}

GlError gl_fri_fold(const uint64_t *d_coeffs, uint64_t len, uint32_t arity_bits, const uint64_t *h_beta, uint64_t *d_out, void *ctx) {
    DeviceCall device_call(ctx);
    if (!ctx || !d_coeffs || !h_beta || !d_out) return fail(GL_E_INVALID, "null pointer");
    hipError_t e = fri_fold(d_coeffs, len, arity_bits, h_beta, d_out, S(ctx)->stream);
    if (e == hipErrorInvalidValue) return fail(GL_E_INVALID, "bad arity / length");
    HIP_TRY(e);
    return ok();
}

GlError gl_ext2_interleave(const uint64_t *d_planes, uint64_t len, uint64_t *d_rows, void *ctx) {
    DeviceCall device_call(ctx);
    if (!ctx || !d_planes || !d_rows) return fail(GL_E_INVALID, "null pointer");
    HIP_TRY(fri_interleave(d_planes, len, d_rows, S(ctx)->stream));
    return ok();
}

GlError gl_fri_proof_of_work(const uint64_t *h_state, uint32_t witness_pos, uint32_t min_leading_zeros, uint64_t *h_witness, void *ctx) {
    DeviceCall device_call(ctx);
    if (!ctx || !h_state || !h_witness) return fail(GL_E_INVALID, "null pointer");
    if (witness_pos >= 8 || min_leading_zeros > 40) return fail(GL_E_INVALID, "bad witness position / difficulty");
    const NttTables *tb;
    HIP_TRY(get_tables(ctx, &tb));
    HIP_TRY(fri_proof_of_work(*tb, h_state, witness_pos, min_leading_zeros, h_witness, S(ctx)->stream));
    return ok();
}

GlError gl_poseidon_permute_batch(uint64_t *d_states, uint64_t count, void *ctx) {
    DeviceCall device_call(ctx);
    if (!ctx || (!d_states && count)) return fail(GL_E_INVALID, "null pointer");
    HIP_TRY(poseidon_permute_batch(d_states, count, S(ctx)->stream));
    return ok();
}

GlError gl_sponge_absorb(uint64_t *h_state, const uint64_t *h_inputs, uint32_t n_blocks, void *ctx) {
    DeviceCall device_call(ctx);
    if (!ctx || !h_state || (!h_inputs && n_blocks)) return fail(GL_E_INVALID, "null pointer");
    const NttTables *tb;
    HIP_TRY(get_tables(ctx, &tb));
    if (12 + 8ull * n_blocks > tb->scratch_elems) return fail(GL_E_INVALID, "too many blocks");
    hipStream_t st = S(ctx)->stream;
    uint64_t *d_state = tb->scratch, *d_in = tb->scratch + 12;
    HIP_TRY(hipMemcpyAsync(d_state, h_state, 96, hipMemcpyHostToDevice, st));
    if (n_blocks) HIP_TRY(hipMemcpyAsync(d_in, h_inputs, 64ull * n_blocks, hipMemcpyHostToDevice, st));
    HIP_TRY(sponge_absorb(d_state, d_in, n_blocks, st));
    HIP_TRY(hipMemcpyAsync(h_state, d_state, 96, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return ok();
}

GlError gl_challenger_step(uint64_t *d_challenger, const GlObserveSrc *h_srcs, uint32_t n_srcs, uint32_t n_challenges, uint64_t *d_out,
                           uint32_t flags, void *ctx) {
    DeviceCall device_call(ctx);
    if (!ctx || !d_challenger || (n_srcs && !h_srcs)) return fail(GL_E_INVALID, "null pointer");
    if (n_srcs > 8) return fail(GL_E_INVALID, "at most eight sources per step");
    if (flags & ~(uint32_t)(GL_CHALLENGER_RESET | GL_CHALLENGER_HASH)) return fail(GL_E_INVALID, "unknown flag");
    if (((flags & GL_CHALLENGER_HASH) || n_challenges) && !d_out) return fail(GL_E_INVALID, "null output");
    const uint64_t *ptrs[8];
    uint64_t counts[8], planar[8];
    for (uint32_t i = 0; i < n_srcs; i++) {
        if (h_srcs[i].count && !h_srcs[i].d_ptr) return fail(GL_E_INVALID, "null source");
        if (h_srcs[i].planar_len && h_srcs[i].count > 2 * h_srcs[i].planar_len) return fail(GL_E_INVALID, "a planar source holds 2 * planar_len elements");
        ptrs[i] = h_srcs[i].d_ptr, counts[i] = h_srcs[i].count, planar[i] = h_srcs[i].planar_len;
    }
    HIP_TRY(challenger_step(d_challenger, ptrs, counts, planar, n_srcs, n_challenges, flags, d_out, S(ctx)->stream));
    return ok();
}

GlError gl_fri_fold_device(const uint64_t *d_coeffs, uint64_t len, uint32_t arity_bits, const uint64_t *d_beta, uint64_t *d_out, void *ctx) {
    DeviceCall device_call(ctx);
    if (!ctx || !d_coeffs || !d_beta || !d_out) return fail(GL_E_INVALID, "null pointer");
    const uint64_t zero[2] = {0, 0};
    hipError_t e = fri_fold(d_coeffs, len, arity_bits, zero, d_out, S(ctx)->stream, d_beta);
    if (e == hipErrorInvalidValue) return fail(GL_E_INVALID, "bad arity / length");
    HIP_TRY(e);
    return ok();
}

GlError gl_fri_proof_of_work_device(const uint64_t *d_challenger, uint32_t min_leading_zeros, uint64_t *d_witness, uint64_t *h_witness, void *ctx) {
    DeviceCall device_call(ctx);
    if (!ctx || !d_challenger || !d_witness || !h_witness) return fail(GL_E_INVALID, "null pointer");
    if (min_leading_zeros > 40) return fail(GL_E_INVALID, "bad difficulty");
    const NttTables *tb;
    HIP_TRY(get_tables(ctx, &tb));
    const uint64_t unused[12] = {0};
    HIP_TRY(fri_proof_of_work(*tb, unused, 0, min_leading_zeros, h_witness, S(ctx)->stream, d_challenger, d_witness));
    return ok();
}

GlError gl_merkle_open_batch_device(const uint64_t *d_leaves, uint64_t row_stride, uint64_t elem_stride, uint32_t leaf_len, uint64_t n_leaves,
                                    uint32_t cap_height, const uint64_t *d_digests, const uint64_t *d_indices, uint32_t count,
                                    uint32_t index_shift, uint64_t *d_out_leaves, uint64_t *d_out_siblings, void *ctx) {
    DeviceCall device_call(ctx);
    if (!ctx || !d_leaves || !d_indices || !d_out_leaves || (!d_out_siblings && (n_leaves >> cap_height) > 1)) return fail(GL_E_INVALID, "null pointer");
    if (n_leaves == 0 || (n_leaves & (n_leaves - 1)) || cap_height > 63 || (1ull << cap_height) > n_leaves || index_shift > 32 ||
        (n_leaves << index_shift) >> index_shift != n_leaves)
        return fail(GL_E_INVALID, "bad tree shape");
    if (count == 0) return ok();
    if ((n_leaves >> cap_height) > 1 && !d_digests) return fail(GL_E_INVALID, "null pointer");
    HIP_TRY(merkle_open_batch(d_leaves, row_stride, elem_stride, leaf_len, n_leaves, cap_height, d_digests, d_indices, count, d_out_leaves,
                              d_out_siblings, S(ctx)->stream, (n_leaves << index_shift) - 1, index_shift));
    return ok();
}

GlError gl_merkle_open_batch(const uint64_t *d_leaves, uint64_t row_stride, uint64_t elem_stride, uint32_t leaf_len, uint64_t n_leaves,
                             uint32_t cap_height, const uint64_t *d_digests, const uint64_t *h_indices, uint32_t count,
                             uint64_t *h_out_leaves, uint64_t *h_out_siblings, void *ctx) {
    DeviceCall device_call(ctx);
    if (!ctx || !d_leaves || !h_indices || !h_out_leaves || (!h_out_siblings && (n_leaves >> cap_height) > 1))
        return fail(GL_E_INVALID, "null pointer");
    if (n_leaves == 0 || (n_leaves & (n_leaves - 1)) || cap_height > 63 || (1ull << cap_height) > n_leaves)
        return fail(GL_E_INVALID, "bad tree shape");
    if (count == 0) return ok();
    for (uint32_t q = 0; q < count; q++)
        if (h_indices[q] >= n_leaves) return fail(GL_E_INVALID, "leaf index out of range");
    uint32_t lg = 0;
    while ((1ull << lg) < n_leaves) lg++;
    const uint64_t layers = lg - cap_height, need = (uint64_t)count * (1 + leaf_len + 4 * layers);
    if (layers && !d_digests) return fail(GL_E_INVALID, "null pointer");
    const NttTables *tb;
    HIP_TRY(get_tables(ctx, &tb));
    if (need > tb->scratch_elems) return fail(GL_E_INVALID, "too many openings for the workspace");
    hipStream_t st = S(ctx)->stream;
    uint64_t *d_idx = tb->scratch, *d_ol = d_idx + count, *d_os = d_ol + (uint64_t)count * leaf_len;
    HIP_TRY(hipMemcpyAsync(d_idx, h_indices, 8ull * count, hipMemcpyHostToDevice, st));
    HIP_TRY(merkle_open_batch(d_leaves, row_stride, elem_stride, leaf_len, n_leaves, cap_height, d_digests, d_idx, count, d_ol, d_os, st));
    HIP_TRY(hipMemcpyAsync(h_out_leaves, d_ol, 8ull * count * leaf_len, hipMemcpyDeviceToHost, st));
    if (layers) HIP_TRY(hipMemcpyAsync(h_out_siblings, d_os, 32ull * count * layers, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return ok();
}

GlError gl_merkle_tree_from_columns(const uint64_t *d_cols, uint32_t leaf_len, uint64_t n_leaves, uint64_t col_stride,
                                    uint32_t cap_height, uint64_t *d_digests, uint64_t *d_cap, void *ctx) {
    DeviceCall device_call(ctx);
    if (!ctx || !d_cols || !d_cap) return fail(GL_E_INVALID, "null pointer");
    if (n_leaves == 0 || (n_leaves & (n_leaves - 1))) return fail(GL_E_INVALID, "n_leaves must be a power of two");
    if ((1ull << cap_height) > n_leaves || cap_height > 63)
        return fail(GL_E_INVALID, "cap_height should be at most log2(leaves.len())");
    HIP_TRY(merkle_tree_from_columns(d_cols, leaf_len, n_leaves, col_stride, cap_height, d_digests, d_cap, S(ctx)->stream));
    return ok();
}

GlError gl_merkle_tree_from_leaves(const uint64_t *d_rows, uint32_t leaf_len, uint64_t n_leaves, uint32_t cap_height,
                                   uint64_t *d_digests, uint64_t *d_cap, void *ctx) {
    DeviceCall device_call(ctx);
    if (!ctx || !d_rows || !d_cap) return fail(GL_E_INVALID, "null pointer");
    if (n_leaves == 0 || (n_leaves & (n_leaves - 1))) return fail(GL_E_INVALID, "n_leaves must be a power of two");
    if ((1ull << cap_height) > n_leaves || cap_height > 63)
        return fail(GL_E_INVALID, "cap_height should be at most log2(leaves.len())");
    HIP_TRY(merkle_tree_from_rows(d_rows, leaf_len, n_leaves, cap_height, d_digests, d_cap, S(ctx)->stream));
    return ok();
}

GlError gl_transpose(const uint64_t *d_cols, uint64_t *d_rows, uint32_t n_cols, uint64_t n_rows, uint64_t col_stride,
                     void *ctx) {
    DeviceCall device_call(ctx);
    if (!ctx || !d_cols || !d_rows) return fail(GL_E_INVALID, "null pointer");
    HIP_TRY(transpose_to_leaf_major(d_cols, d_rows, n_cols, n_rows, col_stride, S(ctx)->stream));
    return ok();
}

GlError gl_pack_leaf_ranges(const uint64_t *d_lde, uint64_t col_stride, uint32_t n_cols, uint64_t leaves_per_rank, uint32_t world,
                            uint64_t *d_out, void *ctx) {
    DeviceCall device_call(ctx);
    if (!ctx || !d_lde || !d_out) return fail(GL_E_INVALID, "null pointer");
    if (world == 0 || n_cols == 0 || leaves_per_rank == 0) return ok();
    if ((leaves_per_rank & 1) || (col_stride & 1) || (((uintptr_t)d_lde | (uintptr_t)d_out) & 15))
        return fail(GL_E_INVALID, "leaf ranges and strides must be even, buffers 16-byte aligned");
    if (col_stride < (uint64_t)world * leaves_per_rank) return fail(GL_E_INVALID, "col_stride smaller than world * leaves_per_rank");
    const uint64_t total = (uint64_t)world * n_cols * (leaves_per_rank / 2);
    const unsigned grid = (unsigned)(total / 256 + 1 > 16384 ? 16384 : total / 256 + 1);
    hipLaunchKernelGGL(pack_leaf_ranges_kernel, dim3(grid), dim3(256), 0, S(ctx)->stream, d_lde, col_stride, n_cols, leaves_per_rank, world, d_out);
    HIP_TRY(hipGetLastError());
    return ok();
}

GlError gl_commit_from_coeffs(const uint64_t *d_coeffs, uint64_t poly_num, uint32_t log_n, uint32_t rate_bits,
                              uint32_t cap_height, uint32_t salt_size, uint64_t shift, uint64_t *d_lde,
                              uint64_t *d_leaves, uint64_t *d_digests, uint64_t *d_cap, void *ctx) {
    DeviceCall device_call(ctx);
    if (log_n > 24) return fail(GL_E_INVALID, "log_n > 24 is not supported by this build");
    return commit_from_coeffs_impl(d_coeffs, poly_num, log_n, rate_bits, cap_height, salt_size, shift, d_lde, d_leaves,
                                   d_digests, d_cap, S(ctx), false);
}

GlError gl_commit_from_values(uint64_t *d_values, uint64_t poly_num, uint32_t log_n, uint32_t rate_bits,
                              uint32_t cap_height, uint32_t salt_size, uint64_t shift, uint64_t *d_lde,
                              uint64_t *d_leaves, uint64_t *d_digests, uint64_t *d_cap, void *ctx) {
    DeviceCall device_call(ctx);
    GlError e = gl_ntt_batch(d_values, poly_num, log_n, 1ull << log_n, 1, 0, ctx);
    if (e.code) return e;
    return gl_commit_from_coeffs(d_values, poly_num, log_n, rate_bits, cap_height, salt_size, shift, d_lde, d_leaves,
                                 d_digests, d_cap, ctx);
}

GlError gl_debug_copy(void *d_dst, const void *d_src, uint64_t bytes, void *ctx) {
    DeviceCall device_call(ctx);
    if (!ctx || !d_dst || !d_src) return fail(GL_E_INVALID, "null pointer");
    if ((bytes & 15) || (((uintptr_t)d_dst | (uintptr_t)d_src) & 15)) return fail(GL_E_INVALID, "16-byte granularity");
    if (bytes == 0) return ok();
    const uint64_t n16 = bytes / 16, per_block = 256ull * COPY_UNROLL;
    hipLaunchKernelGGL(copy16_kernel, dim3((unsigned)((n16 + per_block - 1) / per_block)), dim3(256), 0, S(ctx)->stream,
                       static_cast<const uint64_t *>(d_src), static_cast<uint64_t *>(d_dst), n16);
    HIP_TRY(hipGetLastError());
    return ok();
}

// ops 100 .. 109 of gl_debug_field_op: the register-level radix routines of the NTT passes (ntt_kernels.h) on vectors of sixteen
// elements, one vector per lane — so that the tests can drive their DEFERRED CORRECTION paths with operands that flag (inside a
// transform only the first stage of the first pass ever sees such operands):
//   100 + s (s = 0..3): radix_dif_stage<4, 0, s>      104: radix_dif<4, 0>      105: radix_dif_blocks<2>
//   106 + k (k = 0..3): shift_twiddles_radix4<k>
__global__ void radix_probe_kernel(int which, const uint64_t *in, uint64_t *out, uint64_t n_vec) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t j = i < n_vec ? i : n_vec - 1;   // every lane computes (the masks are per wave); the store is predicated
    uint64_t v[16];
#pragma unroll
    for (int k = 0; k < 16; k++) v[k] = in[j * 16 + k];
    using namespace nttk;
    switch (which) {
        case 0: radix_dif_stage<4, 0, 0>(v); break;
        case 1: radix_dif_stage<4, 0, 1>(v); break;
        case 2: radix_dif_stage<4, 0, 2>(v); break;
        case 3: radix_dif_stage<4, 0, 3>(v); break;
        case 4: radix_dif<4, 0>(v); break;
        case 5: radix_dif_blocks<2>(v); break;
        case 6: shift_twiddles_radix4<0>(v); break;
        case 7: shift_twiddles_radix4<1>(v); break;
        case 8: shift_twiddles_radix4<2>(v); break;
        default: shift_twiddles_radix4<3>(v); break;
    }
    if (i < n_vec)
#pragma unroll
        for (int k = 0; k < 16; k++) out[i * 16 + k] = gl::canon(v[k]);
}

GlError gl_debug_field_op(int op, const uint64_t *d_a, const uint64_t *d_b, uint64_t *d_out, uint64_t n, void *ctx) {
    DeviceCall device_call(ctx);
    if (!ctx || !d_a || !d_out) return fail(GL_E_INVALID, "null pointer");
    if (n == 0) return ok();
    if (op >= 100 && op < 110) {
        if (n % 16) return fail(GL_E_INVALID, "ops 100-109 take vectors of sixteen elements");
        const uint64_t n_vec = n / 16;
        hipLaunchKernelGGL(radix_probe_kernel, dim3((unsigned)((n_vec + 255) / 256)), dim3(256), 0, S(ctx)->stream, op - 100, d_a, d_out, n_vec);
        HIP_TRY(hipGetLastError());
        return ok();
    }
    hipLaunchKernelGGL(field_op_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, S(ctx)->stream, op, d_a, d_b,
                       d_out, n);
    HIP_TRY(hipGetLastError());
    return ok();
}

// ------------------------------------------------------------------------------------------
// reference ABI (cuda/src/lib.rs:58-145)
// ------------------------------------------------------------------------------------------

void init(void) {
    if (hipSetDevice(0) != hipSuccess) return;
    std::lock_guard<std::mutex> lk(g_mu);
    const NttTables *tb;
    (void)device_tables(0, &tb);
}

GlError ifft(uint64_t *d_values_flatten, int poly_num, int values_num_per_poly, int log_len,
             const uint64_t *d_root_table, const uint64_t *n_inv, void *ctx) {
    DeviceCall device_call(ctx);
    (void)d_root_table;
    if (poly_num < 0 || log_len < 0 || values_num_per_poly != (1 << log_len)) return fail(GL_E_INVALID, "bad sizes");
    if (n_inv) {
        uint64_t expect = glh::P - ((glh::P - 1) >> log_len);
        if (*n_inv % glh::P != expect) return fail(GL_E_INVALID, "n_inv does not equal 2^-log_len");
    }
    GlError e = gl_ntt_batch(d_values_flatten, (uint64_t)poly_num, (uint32_t)log_len, (uint64_t)values_num_per_poly, 1, 0, ctx);
    if (e.code) return e;
    HIP_TRY(hipStreamSynchronize(S(ctx)->stream));
    return ok();
}

GlError merkle_tree_from_coeffs(uint64_t *d_values_flatten, uint64_t *d_ext_values_flatten, int poly_num,
                                int values_num_per_poly, int log_len, const uint64_t *d_root_table,
                                const uint64_t *d_root_table2, const uint64_t *d_shift_powers, int rate_bits,
                                int salt_size, int cap_height, int pad_extvalues_len, void *ctx) {
    DeviceCall device_call(ctx);
    (void)d_root_table;
    (void)d_root_table2;
    (void)d_shift_powers;
    if (poly_num <= 0 || log_len < 0 || rate_bits < 0 || salt_size < 0 || cap_height < 0 || pad_extvalues_len < 0 ||
        values_num_per_poly != (1 << log_len))
        return fail(GL_E_INVALID, "bad sizes");
    if (log_len > 24) return fail(GL_E_INVALID, "log_len > 24 is not supported by this build");
    const uint64_t n_ext = (uint64_t)values_num_per_poly << rate_bits;
    const uint64_t ext_polys = (uint64_t)poly_num + salt_size;
    if ((uint64_t)pad_extvalues_len < ext_polys * n_ext)
        return fail(GL_E_INVALID, "pad_extvalues_len smaller than (poly_num+salt_size)*n_ext: regions would overlap");
    uint64_t *region_b = d_ext_values_flatten + pad_extvalues_len;
    uint64_t *digests = region_b + ext_polys * n_ext;
    uint64_t num_digests = 2 * (n_ext - (1ull << cap_height));
    GlError e = commit_from_coeffs_impl(d_values_flatten, (uint64_t)poly_num, (uint32_t)log_len, (uint32_t)rate_bits,
                                        (uint32_t)cap_height, (uint32_t)salt_size, 7, region_b, d_ext_values_flatten,
                                        digests, digests + 4 * num_digests, S(ctx), true);
    if (e.code) return e;
    HIP_TRY(hipStreamSynchronize(S(ctx)->stream));
    // the reference's body ends its hashing with cudaStreamSynchronize(ctx->stream2) (plonky2_gpu.cu:586) and its caller reads the
    // destination of the copy it queued there as soon as this returns (fri/oracle.rs:403-407, 462): stream order already put
    // that copy before the first write of region A; this makes its completion visible to the host as well
    if (!PLONKY2_KNOB("PLONKY2_DROP_STREAM2_WAIT")) HIP_TRY(hipStreamSynchronize(S(ctx)->stream2));
    return ok();
}

GlError merkle_tree_from_values(uint64_t *d_values_flatten, uint64_t *d_ext_values_flatten, int poly_num,
                                int values_num_per_poly, int log_len, const uint64_t *d_root_table,
                                const uint64_t *d_root_table2, const uint64_t *d_shift_powers, const uint64_t *n_inv,
                                int rate_bits, int salt_size, int cap_height, int pad_extvalues_len, void *ctx) {
    DeviceCall device_call(ctx);
    GlError e = ifft(d_values_flatten, poly_num, values_num_per_poly, log_len, d_root_table, n_inv, ctx);
    if (e.code) return e;
    return merkle_tree_from_coeffs(d_values_flatten, d_ext_values_flatten, poly_num, values_num_per_poly, log_len,
                                   d_root_table, d_root_table2, d_shift_powers, rate_bits, salt_size, cap_height,
                                   pad_extvalues_len, ctx);
}

GlError build_merkle_tree(uint64_t *d_ext_values_flatten, int poly_num, int values_num_per_poly, int log_len,
                          int rate_bits, int salt_size, int cap_height, int pad_extvalues_len, void *ctx) {
    DeviceCall device_call(ctx);
    if (!ctx || !d_ext_values_flatten) return fail(GL_E_INVALID, "null pointer");
    if (poly_num <= 0 || log_len < 0 || rate_bits < 0 || salt_size < 0 || cap_height < 0 || pad_extvalues_len < 0 ||
        values_num_per_poly != (1 << log_len) || cap_height > log_len + rate_bits)
        return fail(GL_E_INVALID, "bad sizes");
    const uint32_t log_ext = (uint32_t)(log_len + rate_bits);
    const uint64_t n_ext = 1ull << log_ext, ext_polys = (uint64_t)poly_num + salt_size;
    uint64_t *region_b = d_ext_values_flatten + pad_extvalues_len;
    uint64_t *digests = region_b + ext_polys * n_ext;
    uint64_t num_digests = 2 * (n_ext - (1ull << cap_height));
    uint64_t total = (uint64_t)poly_num * n_ext;  // plonky2_gpu.cu:159-161: salt columns are not permuted
    hipLaunchKernelGGL(bit_reverse_columns_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, S(ctx)->stream,
                       region_b, log_ext, total);
    HIP_TRY(hipGetLastError());
    HIP_TRY(merkle_tree_from_columns(region_b, (uint32_t)ext_polys, n_ext, n_ext, (uint32_t)cap_height, digests,
                                     digests + 4 * num_digests, S(ctx)->stream));
    HIP_TRY(hipStreamSynchronize(S(ctx)->stream));
    return ok();
}

// Column-major staging for compute_quotient_polys: the reference's contract hands over LEAF-MAJOR rows, which the
// quotient kernels read with the row length (1872 B for the wires) as the stride between the lanes of a wave;
// transposing first (streaming, through LDS tiles) and reading column-major is 2x faster end to end. One buffer per
// device — the library's own, grown on demand and given back by gl_reference_quotient_release(), or the CALLER'S
// (gl_reference_quotient_set_staging: a host that sizes all device memory up front, fri/oracle.rs:94-106, keeps doing so and
// the library allocates nothing here). nullptr = could not allocate / the caller's buffer is too small, or
// PLONKY2_HIP_REFERENCE_IN_PLACE=1: read the rows in place. Called with the device's ref_mu held.
static uint64_t *get_ref_staging(DeviceState &st, uint64_t elems) {
    if (const char *v = getenv("PLONKY2_HIP_REFERENCE_IN_PLACE"))
        if (v[0] && v[0] != '0') return nullptr;  // the caller would rather not have the staging buffer
    if (st.ref_staging && !st.ref_staging_owned) return st.ref_staging_elems >= elems ? st.ref_staging : nullptr;
    if (st.ref_staging_elems < elems) {
        if (st.ref_staging) (void)hipFree(st.ref_staging);  // synchronises the device: nothing in flight reads it
        st.ref_staging = nullptr;
        st.ref_staging_elems = 0;
        if (hipMalloc(&st.ref_staging, elems * sizeof(uint64_t)) != hipSuccess) {
            (void)hipGetLastError();  // not an error of the call: fall back to reading the rows in place
            st.ref_staging = nullptr;
            return nullptr;
        }
        st.ref_staging_elems = elems;
        st.ref_staging_owned = true;
    }
    return st.ref_staging;
}

uint64_t gl_reference_quotient_staging_bytes(int log_len) {
    if (log_len < 0 || log_len + (int)ED25519_RATE_BITS > 24) return 0;
    const uint64_t n_ext = (1ull << log_len) << ED25519_RATE_BITS;
    return 8ull * (ED25519_NUM_WIRES + ED25519_CONSTANTS_SIGMAS_LEAF_LEN + ED25519_ZS_PARTIAL_PRODUCTS_LEAF_LEN) * n_ext;
}

GlError gl_reference_quotient_set_staging(void *d_staging, uint64_t bytes) {
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (d_staging && ((uintptr_t)d_staging & 15)) return fail(GL_E_INVALID, "the staging buffer must be 16-byte aligned");
    DeviceState &st = g_dev[dev & 63];
    std::lock_guard<std::mutex> lk(st.ref_mu);  // no compute_quotient_polys is running on this device
    if (st.ref_staging && st.ref_staging_owned) HIP_TRY(hipFree(st.ref_staging));
    st.ref_staging = static_cast<uint64_t *>(d_staging);
    st.ref_staging_elems = d_staging ? bytes / 8 : 0;
    st.ref_staging_owned = false;
    return ok();
}

GlError gl_reference_quotient_release(void) {
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    DeviceState &st = g_dev[dev & 63];
    std::lock_guard<std::mutex> lk(st.ref_mu);
    if (st.ref_staging && st.ref_staging_owned) HIP_TRY(hipFree(st.ref_staging));
    st.ref_staging = nullptr;
    st.ref_staging_elems = 0;
    st.ref_staging_owned = false;
    return ok();
}

GlError gl_reference_quotient_prepare(void *ctx) {
    DeviceCall device_call(ctx);
    if (!ctx) return fail(GL_E_INVALID, "null pointer");
    const GateKernel *k;
    return get_ed25519_kernel(&k);
}

GlError gl_reference_set_public_inputs_hash(const uint64_t *h_hash) {
    std::lock_guard<std::mutex> lk(g_ref_mu);
    for (int k = 0; k < 4; k++) g_ref_pih[k] = h_hash ? h_hash[k] : ED25519_REFERENCE_PUBLIC_INPUTS_HASH[k];
    return ok();
}

GlError gl_reference_set_public_inputs_hash_ctx(const uint64_t *h_hash, void *ctx) {
    DeviceCall device_call(ctx);
    if (!ctx) return fail(GL_E_INVALID, "null ctx");
    CtxState *c;
    HIP_TRY(ctx_state(ctx, &c));
    c->have_pih = h_hash != nullptr;
    for (int k = 0; k < 4; k++) c->pih[k] = h_hash ? h_hash[k] : 0;
    return ok();
}

// cuda/plonky2_gpu.cu:609-783. The circuit (shape, gate table, selector groups) is compiled in, as in the
// reference; what the reference precomputes on the host and hands over as tables (root_table2, shift_inv_powers,
// points, Z_H on the coset and its inverses) the kernels here derive themselves, so those arguments are not read.
GlError compute_quotient_polys(const uint64_t *d_ext_values_flatten, int poly_num, int values_num_per_poly, int log_len,
                               const uint64_t *d_root_table2, const uint64_t *d_shift_inv_powers, int rate_bits, int salt_size,
                               const GlDataSlice *zs_partial_products_commitment_leaves,
                               const GlDataSlice *constants_sigmas_commitment_leaves, void *d_outs, void *d_quotient_polys,
                               const GlDataSlice *points, const GlDataSlice *z_h_on_coset_evals,
                               const GlDataSlice *z_h_on_coset_inverses, const GlDataSlice *k_is, const GlDataSlice *alphas,
                               const GlDataSlice *betas, const GlDataSlice *gammas, void *ctx) {
    DeviceCall device_call(ctx);
    (void)d_root_table2, (void)d_shift_inv_powers, (void)points, (void)z_h_on_coset_evals, (void)z_h_on_coset_inverses;
    const GlDataSlice *zs = zs_partial_products_commitment_leaves, *cs = constants_sigmas_commitment_leaves;
    if (!ctx || !d_ext_values_flatten || !zs || !cs || !d_outs || !d_quotient_polys || !k_is || !alphas || !betas || !gammas)
        return fail(GL_E_INVALID, "null pointer");
    if (!zs->ptr || !cs->ptr || !k_is->ptr || !alphas->ptr || !betas->ptr || !gammas->ptr) return fail(GL_E_INVALID, "null pointer in DataSlice");
    if (log_len < 0 || log_len + (int)ED25519_RATE_BITS > 24 || values_num_per_poly != (1 << log_len)) return fail(GL_E_INVALID, "bad sizes");
    if (salt_size < 0 || poly_num != (int)ED25519_NUM_WIRES || rate_bits != (int)ED25519_RATE_BITS)
        return fail(GL_E_INVALID, "compute_quotient_polys is compiled for the ed25519 circuit: 234 wire polynomials, rate_bits 3 "
                                  "(plonky2_gpu.cu:666-675); use gl_compute_quotient_polys for any other circuit");
    const uint64_t n_ext = (uint64_t)values_num_per_poly << rate_bits;
    // the reference's own asserts (plonky2_gpu.cu:677-683)
    if ((uint64_t)cs->len != n_ext * ED25519_CONSTANTS_SIGMAS_LEAF_LEN || (uint64_t)zs->len != n_ext * ED25519_ZS_PARTIAL_PRODUCTS_LEAF_LEN)
        return fail(GL_E_INVALID, "leaf buffers must hold n_ext x 88 (constants_sigmas) and n_ext x 20 (zs_partial_products) elements");
    if (alphas->len != (int)ED25519_NUM_CHALLENGES || betas->len != (int)ED25519_NUM_CHALLENGES || gammas->len != (int)ED25519_NUM_CHALLENGES)
        return fail(GL_E_INVALID, "alphas, betas and gammas must hold num_challenges = 2 elements each");
    if (k_is->len < (int)ED25519_NUM_ROUTED_WIRES) return fail(GL_E_INVALID, "k_is must hold num_routed_wires = 80 elements");
    const GateKernel *kernel;
    GlError ge = get_ed25519_kernel(&kernel);
    if (ge.code) return ge;
    // the challenges live in device memory on the reference's side of the boundary (prover.rs:489-516)
    uint64_t ch[3][2];
    const GlDataSlice *src[3] = {alphas, betas, gammas};
    for (int i = 0; i < 3; i++) HIP_TRY(hipMemcpyAsync(ch[i], src[i]->ptr, sizeof(ch[i]), hipMemcpyDeviceToHost, S(ctx)->stream));
    HIP_TRY(hipStreamSynchronize(S(ctx)->stream));
    // The circuit's public-inputs hash has no slot in the reference's signature (its kernel has the proof's value compiled in,
    // plonky2_gpu_impl.cuh:600-685): the context's own (gl_reference_set_public_inputs_hash_ctx) if it has one, else the process's.
    uint64_t pih[4];
    CtxState *cst;
    HIP_TRY(ctx_state(ctx, &cst));
    if (cst->have_pih) {
        for (int k = 0; k < 4; k++) pih[k] = cst->pih[k];
    } else {
        std::lock_guard<std::mutex> lk(g_ref_mu);
        for (int k = 0; k < 4; k++) pih[k] = g_ref_pih[k];
    }
    // one gate-kernel object (its constant tables) and one staging buffer per device: calls of this symbol on one device take
    // turns, held to the stream synchronisation that ends the call
    DeviceState &dst = g_dev[cst->dev & 63];
    std::lock_guard<std::mutex> ref_turn(dst.ref_mu);
    GlQuotientArgs a = {};
    a.d_wires_leaves = d_ext_values_flatten;  // leaf-major, leaf t = point bitrev(t) (plonky2_gpu_impl.cuh:537-541)
    a.d_constants_sigmas_leaves = static_cast<const uint64_t *>(cs->ptr);
    a.d_zs_partial_products_leaves = static_cast<const uint64_t *>(zs->ptr);
    a.wires_leaf_len = (uint32_t)(poly_num + salt_size);
    a.constants_sigmas_leaf_len = ED25519_CONSTANTS_SIGMAS_LEAF_LEN;
    a.zs_partial_products_leaf_len = ED25519_ZS_PARTIAL_PRODUCTS_LEAF_LEN;
    a.d_k_is = static_cast<const uint64_t *>(k_is->ptr);
    a.h_alphas = ch[0], a.h_betas = ch[1], a.h_gammas = ch[2];
    a.num_constants = ED25519_NUM_CONSTANTS;
    a.num_routed_wires = ED25519_NUM_ROUTED_WIRES;
    a.num_challenges = ED25519_NUM_CHALLENGES;
    a.num_gate_constraints = ED25519_NUM_GATE_CONSTRAINTS;
    a.degree_bits = (uint32_t)log_len;
    a.rate_bits = (uint32_t)rate_bits;
    a.quotient_degree_factor = ED25519_QUOTIENT_DEGREE_FACTOR;
    a.coset_shift = 7;
    a.column_stride = 0;
    if (uint64_t *stage = get_ref_staging(dst, (uint64_t)(a.wires_leaf_len + a.constants_sigmas_leaf_len + a.zs_partial_products_leaf_len) * n_ext)) {
        uint64_t *w = stage, *c = w + (uint64_t)a.wires_leaf_len * n_ext, *z = c + (uint64_t)a.constants_sigmas_leaf_len * n_ext;
        HIP_TRY(transpose_to_column_major(a.d_wires_leaves, w, a.wires_leaf_len, n_ext, n_ext, S(ctx)->stream));
        HIP_TRY(transpose_to_column_major(a.d_constants_sigmas_leaves, c, a.constants_sigmas_leaf_len, n_ext, n_ext, S(ctx)->stream));
        HIP_TRY(transpose_to_column_major(a.d_zs_partial_products_leaves, z, a.zs_partial_products_leaf_len, n_ext, n_ext, S(ctx)->stream));
        a.d_wires_leaves = w, a.d_constants_sigmas_leaves = c, a.d_zs_partial_products_leaves = z;
        a.column_stride = n_ext;
    }
    a.gate_kernel = kernel;
    a.h_public_inputs_hash = pih;
    a.d_gate_workspace = static_cast<uint64_t *>(d_outs);  // [2][n_ext]: the reference's scratch for the same stage
    GlError e = gl_compute_quotient_polys(&a, static_cast<uint64_t *>(d_quotient_polys), ctx);
    if (e.code) return e;
    HIP_TRY(hipStreamSynchronize(S(ctx)->stream));
    return ok();
}

const char *cudaGetErrorString(int code) {
    if (code < 0) return code == GL_E_INVALID ? "plonky2_hip: invalid argument" : "plonky2_hip: unsupported";
    return hipGetErrorString((hipError_t)code);
}

}  // extern "C"
