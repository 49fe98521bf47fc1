// ubench_mem.hip — what the memory system gives the NTT's access patterns on gfx950, with no arithmetic and no LDS:
// every "tile" is loaded into registers with the pattern of one pass's load and stored with the pattern of its store.
// Sizes the skeleton of csrc/ntt.hip: which segment width, tile size and chunking the passes should use.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_mem.hip -o tools/ubench_mem
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)

struct alignas(16) v2 { uint64_t x, y; };

// A pattern: the c-th 16-byte piece of a tile lives at  base + (c >> logw) * stride + (c & (2^logw - 1)) * 2   (elements)
struct Pat {
    uint64_t col_stride;   // blockIdx.y
    uint64_t tile_stride;  // blockIdx.x
    uint64_t seg_stride;
    uint32_t logw;         // log2 of 16-byte pieces per contiguous segment
};

template <int NT, int ITER, bool NT_HINT>
__global__ __launch_bounds__(NT) void tile_kernel(const uint64_t *__restrict__ src, uint64_t *__restrict__ dst, Pat in, Pat out) {
    const uint32_t tid = threadIdx.x;
    const uint64_t ib = blockIdx.y * in.col_stride + blockIdx.x * in.tile_stride;
    const uint64_t ob = blockIdx.y * out.col_stride + blockIdx.x * out.tile_stride;
    v2 v[ITER];
#pragma unroll
    for (int it = 0; it < ITER; it++) {
        uint32_t c = tid + it * NT;
        const v2 *p = reinterpret_cast<const v2 *>(src + ib + (uint64_t)(c >> in.logw) * in.seg_stride + (c & ((1u << in.logw) - 1)) * 2);
        if constexpr (NT_HINT) {
            v[it].x = __builtin_nontemporal_load(&p->x);
            v[it].y = __builtin_nontemporal_load(&p->y);
        } else {
            v[it] = *p;
        }
    }
#pragma unroll
    for (int it = 0; it < ITER; it++) {
        uint32_t c = tid + it * NT;
        v2 *p = reinterpret_cast<v2 *>(dst + ob + (uint64_t)(c >> out.logw) * out.seg_stride + (c & ((1u << out.logw) - 1)) * 2);
        v2 val = v[it];
        val.x ^= 1;
        if constexpr (NT_HINT) {
            __builtin_nontemporal_store(val.x, &p->x);
            __builtin_nontemporal_store(val.y, &p->y);
        } else {
            *p = val;
        }
    }
}

// 8 bytes per lane: lane-contiguous dwordx2 (what a wave that loads straight into its radix-16 operands would issue)
template <int NT, int ITER>
__global__ __launch_bounds__(NT) void tile8_kernel(const uint64_t *__restrict__ src, uint64_t *__restrict__ dst, uint64_t col_stride, uint64_t tile_stride) {
    const uint64_t b = blockIdx.y * col_stride + blockIdx.x * tile_stride;
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint64_t v[ITER];
    const uint64_t wb = b + (uint64_t)wave * 64 * ITER;
#pragma unroll
    for (int it = 0; it < ITER; it++) v[it] = src[wb + it * 64 + lane];
#pragma unroll
    for (int it = 0; it < ITER; it++) dst[wb + it * 64 + lane] = v[it] ^ 1;
}

__global__ __launch_bounds__(256) void copy_gs(const v2 *__restrict__ in, v2 *__restrict__ out, uint64_t n) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) out[i] = in[i];
}

template <int U>
__global__ __launch_bounds__(256) void copy_unroll(const v2 *__restrict__ in, v2 *__restrict__ out, uint64_t n) {
    uint64_t i = ((uint64_t)blockIdx.x * U) * 256 + threadIdx.x;
    v2 v[U];
#pragma unroll
    for (int u = 0; u < U; u++) v[u] = in[i + u * 256];
#pragma unroll
    for (int u = 0; u < U; u++) out[i + u * 256] = v[u];
}

__global__ __launch_bounds__(256) void fill_kernel(uint64_t *p, uint64_t n) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) p[i] = i * 0x9E3779B97F4A7C15ull;
}

static hipEvent_t e0, e1;
template <class F>
static double time_ms(F &&f, int reps = 7) {
    std::vector<float> t;
    for (int r = 0; r < reps; r++) {
        CK(hipEventRecord(e0));
        f();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (r) t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
}

int main() {
    const uint64_t LOGN = 20, N = 1ull << LOGN, COLS = 64, TOTAL = COLS * N;
    uint64_t *a, *b, *mid;
    CK(hipMalloc(&a, (TOTAL + COLS * 1024 * 512) * 8));  // room for padded row strides
    CK(hipMalloc(&b, (TOTAL + COLS * 1024 * 512) * 8));
    CK(hipMalloc(&mid, 16 * N * 8));
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, a, TOTAL);
    hipLaunchKernelGGL(fill_kernel, dim3(4096), dim3(256), 0, 0, b, TOTAL);
    CK(hipDeviceSynchronize());
    const double GB = 2.0 * TOTAL * 8 / 1e9;  // read + write of the 64-column batch
    auto report = [&](const char *name, double ms, double gb) { printf("%-78s %8.3f ms  %7.1f GB/s\n", name, ms, gb / (ms * 1e-3)); fflush(stdout); };

    // ---- plain copies ---------------------------------------------------------------------------------------------
    for (int blocks : {2048, 4096, 8192, 16384})
        { char nm[128]; snprintf(nm, sizeof nm, "copy grid-stride 16B/lane, %d blocks x 256", blocks);
          report(nm, time_ms([&] { hipLaunchKernelGGL(copy_gs, dim3(blocks), dim3(256), 0, 0, (const v2 *)a, (v2 *)b, TOTAL / 2); }), GB); }
    report("copy one-shot unroll 4 (16 KiB per block)", time_ms([&] { hipLaunchKernelGGL(copy_unroll<4>, dim3((unsigned)(TOTAL / 2 / 256 / 4)), dim3(256), 0, 0, (const v2 *)a, (v2 *)b, TOTAL / 2); }), GB);
    report("copy one-shot unroll 8 (32 KiB per block)", time_ms([&] { hipLaunchKernelGGL(copy_unroll<8>, dim3((unsigned)(TOTAL / 2 / 256 / 8)), dim3(256), 0, 0, (const v2 *)a, (v2 *)b, TOTAL / 2); }), GB);
    report("copy one-shot unroll 16 (64 KiB per block)", time_ms([&] { hipLaunchKernelGGL(copy_unroll<16>, dim3((unsigned)(TOTAL / 2 / 256 / 16)), dim3(256), 0, 0, (const v2 *)a, (v2 *)b, TOTAL / 2); }), GB);

    // ---- tile patterns (64 columns in one launch, a -> b) -------------------------------------------------------------
    // contiguous tile of E elements
    auto contig = [&](uint64_t E) { return Pat{N, E, 0, 31}; };
    // column pass of an R x (N/R) matrix: tile = R rows x T columns, T = E/R; segment = T elements at stride N/R
    auto colpat = [&](uint64_t E, uint64_t R) { uint64_t T = E / R; uint32_t lw = 0; while ((2ull << lw) < T) lw++; return Pat{N, T, N / R, T >= 2 ? lw : 0}; };
    // the same with the rows of the R x (N/R) matrix pad elements apart (is the 8 KiB stride itself a problem for the channels?)
    auto colpad = [&](uint64_t E, uint64_t R, uint64_t pad) { Pat q = colpat(E, R); q.col_stride = N + R * pad; q.seg_stride = N / R + pad; return q; };
    struct Case { const char *name; int nt, iter; uint64_t E; Pat in, out; };
    std::vector<Case> cases = {
        {"tile 64KiB/512thr: contiguous -> contiguous (row pass, in place)", 512, 8, 8192, contig(8192), contig(8192)},
        {"tile 64KiB/512thr: 64B segs stride 8KiB -> same (column pass R=1024, now)", 512, 8, 8192, colpat(8192, 1024), colpat(8192, 1024)},
        {"tile 64KiB/512thr: 128B segs stride 16KiB -> same (column pass R=512)", 512, 8, 8192, colpat(8192, 512), colpat(8192, 512)},
        {"tile 64KiB/512thr: 256B segs stride 32KiB -> same (column pass R=256)", 512, 8, 8192, colpat(8192, 256), colpat(8192, 256)},
        {"tile 64KiB/512thr: 32B segs stride 4KiB -> same (column pass R=2048)", 512, 8, 8192, colpat(8192, 2048), colpat(8192, 2048)},
        {"tile 64KiB/512thr: contiguous -> 64B segs stride 8KiB (row pass, natural out)", 512, 8, 8192, contig(8192), colpat(8192, 1024)},
        {"tile 64KiB/512thr: 64B segs -> contiguous", 512, 8, 8192, colpat(8192, 1024), contig(8192)},
        {"tile 128KiB/1024thr: contiguous -> contiguous", 1024, 8, 16384, contig(16384), contig(16384)},
        {"tile 128KiB/1024thr: 128B segs stride 8KiB -> same (column pass R=1024, T=16)", 1024, 8, 16384, colpat(16384, 1024), colpat(16384, 1024)},
        {"tile 128KiB/1024thr: contiguous -> 128B segs stride 8KiB (row pass natural, T=16)", 1024, 8, 16384, contig(16384), colpat(16384, 1024)},
        {"tile 128KiB/512thr x16: 128B segs stride 8KiB -> same", 512, 16, 16384, colpat(16384, 1024), colpat(16384, 1024)},
        {"tile 128KiB/1024thr: 128B segs stride 8KiB+128B -> same", 1024, 8, 16384, colpad(16384, 1024, 16), colpad(16384, 1024, 16)},
        {"tile 128KiB/1024thr: 128B segs stride 8KiB+256B -> same", 1024, 8, 16384, colpad(16384, 1024, 32), colpad(16384, 1024, 32)},
        {"tile 128KiB/1024thr: 128B segs stride 8KiB+1152B -> same", 1024, 8, 16384, colpad(16384, 1024, 144), colpad(16384, 1024, 144)},
        {"tile 64KiB/512thr: 64B segs stride 8KiB+64B -> same", 512, 8, 8192, colpad(8192, 1024, 8), colpad(8192, 1024, 8)},
        {"tile 64KiB/512thr: 64B segs stride 8KiB+192B -> same", 512, 8, 8192, colpad(8192, 1024, 24), colpad(8192, 1024, 24)},
        {"tile 64KiB/512thr: contiguous -> 64B segs stride 8KiB+192B", 512, 8, 8192, contig(8192), colpad(8192, 1024, 24)},
        {"tile 32KiB/256thr: contiguous -> contiguous", 256, 8, 4096, contig(4096), contig(4096)},
        {"tile 32KiB/256thr: 64B segs stride 16KiB -> same (column pass R=512, T=8)", 256, 8, 4096, colpat(4096, 512), colpat(4096, 512)},
    };
    auto launch = [&](const Case &c, const uint64_t *src, uint64_t *dst, unsigned cols, bool nt_hint) {
        dim3 grid((unsigned)(N / c.E), cols);
#define L(NTH, IT) \
        if (c.nt == NTH && c.iter == IT) { \
            if (nt_hint) hipLaunchKernelGGL((tile_kernel<NTH, IT, true>), grid, dim3(NTH), 0, 0, src, dst, c.in, c.out); \
            else hipLaunchKernelGGL((tile_kernel<NTH, IT, false>), grid, dim3(NTH), 0, 0, src, dst, c.in, c.out); }
        L(512, 8) L(1024, 8) L(512, 16) L(256, 8)
#undef L
    };
    // every case must stay inside the allocations: the largest element offset a launch touches, from its pattern
    const uint64_t ALLOC = TOTAL + COLS * 1024 * 512;
    auto max_offset = [&](const Case &c, const Pat &q, unsigned cols) {
        const uint64_t pieces = c.E / 2, last = pieces - 1;
        return (uint64_t)(cols - 1) * q.col_stride + (N / c.E - 1) * q.tile_stride + (last >> q.logw) * q.seg_stride + (last & ((1ull << q.logw) - 1)) * 2 + 1;
    };
    auto in_bounds = [&](const Case &c, unsigned cols) {
        const uint64_t mi = max_offset(c, c.in, cols), mo = max_offset(c, c.out, cols);
        if (mi >= ALLOC || mo >= ALLOC) {
            printf("%-78s SKIPPED: would touch element %llu of %llu\n", c.name, (unsigned long long)(mi > mo ? mi : mo), (unsigned long long)ALLOC);
            return false;
        }
        return true;
    };
    for (auto &c : cases)
        if (in_bounds(c, COLS)) report(c.name, time_ms([&] { launch(c, a, b, COLS, false); }), GB);
    for (int k : {0, 1, 5}) { char nm[160]; snprintf(nm, sizeof nm, "[nontemporal] %s", cases[k].name); report(nm, time_ms([&] { launch(cases[k], a, b, COLS, true); }), GB); }
    report("tile 64KiB/512thr: 8B per lane, lane-contiguous (16 x dwordx2 per thread)", time_ms([&] { hipLaunchKernelGGL((tile8_kernel<512, 16>), dim3((unsigned)(N / 8192), COLS), dim3(512), 0, 0, a, b, N, (uint64_t)8192); }), GB);
    report("tile 8KiB/64thr: 8B per lane (one wave = one 1024-point row, 16 x dwordx2)", time_ms([&] { hipLaunchKernelGGL((tile8_kernel<64, 16>), dim3((unsigned)(N / 1024), COLS), dim3(64), 0, 0, a, b, N, (uint64_t)1024); }), GB);

    // ---- two passes with the intermediate in a scratch buffer, chunked by columns (Infinity Cache residency) --------------
    // pass A: a -> mid (pattern pa both sides), pass B: mid -> b (contiguous in, pattern pb out), per chunk of C columns
    struct Two { const char *name; int ia, ib; };
    std::vector<Two> twos = {{"2 passes 64KiB tiles: A = 64B segs in/out, B = contiguous -> 64B segs", 1, 5},
                             {"2 passes 128KiB tiles: A = 128B segs in/out, B = contiguous -> 128B segs", 8, 9},
                             {"2 passes: A = contiguous, B = contiguous (upper bound)", 0, 0}};
    for (auto &t : twos)
        for (unsigned C : {1u, 2u, 4u, 8u, 16u}) {
            char nm[200];
            snprintf(nm, sizeof nm, "%s, chunk %u cols", t.name, C);
            double ms = time_ms([&] {
                for (unsigned off = 0; off < COLS; off += C) {
                    launch(cases[t.ia], a + off * N, mid, C, false);
                    launch(cases[t.ib], mid, b + off * N, C, false);
                }
            }, 5);
            report(nm, ms, 2 * GB);
        }
    // in place, all columns per pass (the LDE / commit path today)
    report("2 passes in place, all 64 columns per pass: A 64B segs, B contiguous", time_ms([&] { launch(cases[1], b, b, COLS, false); launch(cases[0], b, b, COLS, false); }, 5), 2 * GB);
    for (unsigned C : {2u, 4u, 8u, 16u}) {
        char nm[200];
        snprintf(nm, sizeof nm, "2 passes in place, chunk %u cols: A 64B segs, B contiguous", C);
        report(nm, time_ms([&] { for (unsigned off = 0; off < COLS; off += C) { launch(cases[1], b + off * N, b + off * N, C, false); launch(cases[0], b + off * N, b + off * N, C, false); } }, 5), 2 * GB);
    }
    return 0;
}
