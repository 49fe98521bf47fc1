// Experiment / micro-benchmark: the Poseidon permutation with its MDS layers on the matrix cores (csrc/poseidon.h) against
// the vector-ALU permutation (csrc/poseidon_vector.h): bit-equality on random and edge states, and time per 2^22 permutations
// for both, and the MDS layer alone. (profiles/r03_poseidon_matrix_cores.jsonl also has the variant that was not kept: matrix-core
// full rounds around the vector ALU's blocked partial rounds.)
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form -I plonky2_gpu_amd/csrc tools/experiments/mds_mfma.hip -o mds_mfma
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#include "poseidon.h"

template <int WHICH>
__global__ __launch_bounds__(256) void permute_kernel(uint64_t *states, int reps) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const poseidon::MdsOperands ops = poseidon::mds_operands();
    uint64_t s[12];
#pragma unroll
    for (int k = 0; k < 12; k++) s[k] = states[i * 12 + k];
#pragma unroll 1
    for (int l = 0; l < reps; l++) {
        if constexpr (WHICH == 0) poseidon_vector::permute(s);
        else poseidon::permute(s, ops);
    }
#pragma unroll
    for (int k = 0; k < 12; k++) states[i * 12 + k] = gl::canon(s[k]);
}

// the MDS layer alone (with the constants that follow the full rounds), vector against matrix cores
template <bool MFMA>
__global__ __launch_bounds__(256) void layers_kernel(uint64_t *states, int n_layers) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const poseidon::MdsOperands ops = poseidon::mds_operands();
    uint64_t s[12];
#pragma unroll
    for (int k = 0; k < 12; k++) s[k] = states[i * 12 + k];
#pragma unroll 1
    for (int l = 0; l < n_layers; l++) {
        const int q = l % 6;  // the constants after rounds 0-2 and 25-27
        if constexpr (MFMA) poseidon::mds_layer(s, ops, POSEIDON_MDS_XY + 24 * (q < 3 ? q : q + 22));
        else poseidon_vector::mds_layer(s, POSEIDON_ALL_ROUND_CONSTANTS + 12 * ((q < 3 ? q : q + 22) + 1));
    }
#pragma unroll
    for (int k = 0; k < 12; k++) states[i * 12 + k] = gl::canon(s[k]);
}

int main() {
    const uint64_t n = 1 << 22;
    std::vector<uint64_t> h(n * 12);
    uint64_t x = 88172645463325252ull;
    for (auto &v : h) {
        x ^= x << 13, x ^= x >> 7, x ^= x << 17;
        v = x;  // any 64-bit value, canonical or not
    }
    for (int k = 0; k < 12; k++)
        h[k] = 0xFFFFFFFFFFFFFFFFull, h[12 + k] = 0, h[24 + k] = 0xFFFFFFFF00000000ull, h[36 + k] = 0x8080808080808080ull, h[48 + k] = 0x7F7F7F7F7F7F7F7Full;
    uint64_t *d[2];
    for (auto &p : d)
        if (hipMalloc(&p, n * 96) != hipSuccess) return 2;
    std::vector<uint64_t> out[2];
    for (auto &o : out) o.resize(n * 12);
    hipEvent_t e[3];
    for (auto &ev : e) (void)hipEventCreate(&ev);
    for (int layers : {1, 64}) {
        for (int w = 0; w < 2; w++) (void)hipMemcpy(d[w], h.data(), n * 96, hipMemcpyHostToDevice);
        (void)hipEventRecord(e[0]);
        hipLaunchKernelGGL(layers_kernel<false>, dim3(n / 256), dim3(256), 0, 0, d[0], layers);
        (void)hipEventRecord(e[1]);
        hipLaunchKernelGGL(layers_kernel<true>, dim3(n / 256), dim3(256), 0, 0, d[1], layers);
        (void)hipEventRecord(e[2]);
        (void)hipDeviceSynchronize();
        float t0, t1;
        (void)hipEventElapsedTime(&t0, e[0], e[1]), (void)hipEventElapsedTime(&t1, e[1], e[2]);
        for (int w = 0; w < 2; w++) (void)hipMemcpy(out[w].data(), d[w], n * 96, hipMemcpyDeviceToHost);
        uint64_t bad = 0;
        for (uint64_t k = 0; k < n * 12; k++) bad += out[0][k] != out[1][k];
        printf("{\"mds_layers\": %d, \"states\": %llu, \"vector_ms\": %.3f, \"matrix_core_ms\": %.3f, \"mismatching_words\": %llu}\n", layers,
               (unsigned long long)n, t0, t1, (unsigned long long)bad);
    }
    for (int reps : {1, 4, 4}) {
        for (int w = 0; w < 2; w++) (void)hipMemcpy(d[w], h.data(), n * 96, hipMemcpyHostToDevice);
        (void)hipEventRecord(e[0]);
        hipLaunchKernelGGL(permute_kernel<0>, dim3(n / 256), dim3(256), 0, 0, d[0], reps);
        (void)hipEventRecord(e[1]);
        hipLaunchKernelGGL(permute_kernel<1>, dim3(n / 256), dim3(256), 0, 0, d[1], reps);
        (void)hipEventRecord(e[2]);
        (void)hipDeviceSynchronize();
        float t[2];
        for (int w = 0; w < 2; w++) (void)hipEventElapsedTime(&t[w], e[w], e[w + 1]);
        for (int w = 0; w < 2; w++) (void)hipMemcpy(out[w].data(), d[w], n * 96, hipMemcpyDeviceToHost);
        uint64_t bad = 0;
        for (uint64_t k = 0; k < n * 12; k++) bad += out[0][k] != out[1][k];
        printf("{\"permutations_per_state\": %d, \"states\": %llu, \"vector_ms\": %.3f, \"matrix_core_ms\": %.3f, \"G_perm_per_s\": [%.3f, %.3f], "
               "\"mismatching_words\": %llu}\n",
               reps, (unsigned long long)n, t[0], t[1], n * reps / t[0] / 1e6, n * reps / t[1] / 1e6, (unsigned long long)bad);
    }
    return 0;
}
