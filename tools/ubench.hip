// ubench.hip — instruction-rate microbenchmarks on gfx950 that size the kernels' ALU roofs.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I plonky2_gpu_amd/csrc tools/ubench.hip -o tools/ubench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#include "poseidon.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

constexpr int ITERS = 2048;

template <int OP>
__global__ __launch_bounds__(256) void alu_kernel(uint64_t *out, uint64_t seed) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t a[4];
#pragma unroll
    for (int k = 0; k < 4; k++) a[k] = seed * (i + 1) + k * 0x9E3779B97F4A7C15ull;
    uint64_t b = gl::canon(seed ^ 0xD1B54A32D192ED03ull);
#pragma unroll
    for (int k = 0; k < 4; k++) a[k] = gl::canon(a[k]);
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if constexpr (OP == 0) a[k] = gl::mul(a[k], b);
            if constexpr (OP == 1) a[k] = gl::add(a[k], b);
            if constexpr (OP == 2) a[k] = gl::sub(a[k], b);
            if constexpr (OP == 3) a[k] = gl::mul_pow2<39>(a[k]);
            if constexpr (OP == 4) a[k] = gl::mul_pow2<156>(a[k]);
            if constexpr (OP == 5) a[k] = gl::sqr(a[k]);
            if constexpr (OP == 6) a[k] = gl::mac(a[k], a[(k + 1) & 3], b);
            if constexpr (OP == 7) { uint64_t lo, hi; gl::mul_wide(a[k], b, lo, hi); a[k] = lo ^ hi; }
            if constexpr (OP == 8) a[k] = a[k] * 0x9E3779B97F4A7C15ull + b;  // plain 64-bit mul low
            if constexpr (OP == 9) a[k] = (uint64_t)((uint32_t)a[k]) * (uint32_t)b + a[k];  // one v_mad_u64_u32
            if constexpr (OP == 10) a[k] = gl::add_c(a[k], b);
            if constexpr (OP == 11) a[k] = gl::sub_c(a[k], b);
            if constexpr (OP == 12) a[k] = gl::mul_c(a[k], b);
            if constexpr (OP == 13) a[k] = gl::canon_c(a[k] + b);
        }
    }
    out[i] = a[0] ^ a[1] ^ a[2] ^ a[3];
}

#ifdef POSEIDON_WAVES
__attribute__((amdgpu_waves_per_eu(POSEIDON_WAVES, POSEIDON_WAVES)))
#endif
__global__ __launch_bounds__(256) void poseidon_kernel(uint64_t *out, uint64_t seed, int perms) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t s[12];
#pragma unroll
    for (int k = 0; k < 12; k++) s[k] = seed * (i + 1) + k;
    for (int p = 0; p < perms; p++) poseidon::permute(s);
    out[i] = s[0] ^ s[5];
}

struct alignas(16) v2 { uint64_t x, y; };
__global__ __launch_bounds__(256) void copy_kernel(const v2 *in, v2 *out, uint64_t n) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) out[i] = in[i];
}

template <class F>
float time_ms(F f, int reps = 5) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f();
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < reps; r++) {
        CK(hipEventRecord(a));
        f();
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    return best;
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("device: %s, CUs %d, clock %d MHz\n", prop.name, prop.multiProcessorCount, prop.clockRate / 1000);
    const int blocks = prop.multiProcessorCount * 8, threads = 256;
    uint64_t *out;
    CK(hipMalloc(&out, (size_t)blocks * threads * 8));
    const char *names[] = {"gl::mul", "gl::add", "gl::sub", "mul_pow2<39>", "mul_pow2<156>", "gl::sqr", "gl::mac",
                           "mul_wide(64x64->128)", "u64 mul lo", "v_mad_u64_u32",
                           "add_c (asm)", "sub_c (asm)", "mul_c (asm reduce)", "canon_c (asm)+add64"};
    auto run = [&](auto tag, int op) {
        constexpr int OP = decltype(tag)::value;
        float ms = time_ms([&] { hipLaunchKernelGGL(alu_kernel<OP>, dim3(blocks), dim3(threads), 0, 0, out, 12345ull); });
        double ops = (double)blocks * threads * ITERS * 4;
        printf("%-22s %8.3f ms  %8.2f Gop/s  (%.1f lane-cycles/op at %d MHz, %d lanes)\n", names[op], ms, ops / ms / 1e6,
               (double)prop.multiProcessorCount * 128 * (prop.clockRate * 1e3) / (ops / (ms * 1e-3)), prop.clockRate / 1000,
               prop.multiProcessorCount * 128);
    };
    run(std::integral_constant<int, 0>{}, 0);
    run(std::integral_constant<int, 1>{}, 1);
    run(std::integral_constant<int, 2>{}, 2);
    run(std::integral_constant<int, 3>{}, 3);
    run(std::integral_constant<int, 4>{}, 4);
    run(std::integral_constant<int, 5>{}, 5);
    run(std::integral_constant<int, 6>{}, 6);
    run(std::integral_constant<int, 7>{}, 7);
    run(std::integral_constant<int, 8>{}, 8);
    run(std::integral_constant<int, 9>{}, 9);
    run(std::integral_constant<int, 10>{}, 10);
    run(std::integral_constant<int, 11>{}, 11);
    run(std::integral_constant<int, 12>{}, 12);
    run(std::integral_constant<int, 13>{}, 13);
    {
        int perms = 64;
        float ms = time_ms([&] { hipLaunchKernelGGL(poseidon_kernel, dim3(blocks), dim3(threads), 0, 0, out, 777ull, perms); });
        double n = (double)blocks * threads * perms;
        printf("poseidon permute       %8.3f ms  %8.2f Mperm/s\n", ms, n / ms / 1e3);
    }
    {
        size_t bytes = 1ull << 30;
        v2 *a, *b;
        CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
        CK(hipMemset(a, 1, bytes));
        float ms = time_ms([&] { hipLaunchKernelGGL(copy_kernel, dim3(blocks), dim3(threads), 0, 0, a, b, bytes / 16); });
        printf("copy 1 GiB (16B/lane)  %8.3f ms  %8.1f GB/s (read+write)\n", ms, 2.0 * bytes / ms / 1e6);
    }
    return 0;
}
