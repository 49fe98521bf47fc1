// Experiment / micro-benchmark: the Poseidon permutation with its MDS layers on the matrix cores (csrc/poseidon.h) against
// the vector-ALU permutation (csrc/poseidon_vector.h): bit-equality on random and edge states, and time per 2^22 permutations
// for both, and the MDS layer alone. (profiles/r03_poseidon_matrix_cores.jsonl also has the variant that was not kept: matrix-core
// full rounds around the vector ALU's blocked partial rounds.)
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form -I plonky2_gpu_amd/csrc tools/experiments/mds_mfma.hip -o mds_mfma
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#include "poseidon_blocked.h"

#ifndef BLOCK_WAVES
#define BLOCK_WAVES 2
#endif
#ifndef PLAIN_WAVES
#define PLAIN_WAVES 4
#endif
template <int WHICH>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WHICH == 2 ? BLOCK_WAVES : PLAIN_WAVES, WHICH == 2 ? BLOCK_WAVES : PLAIN_WAVES))) void permute_kernel(uint64_t *states, int reps) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const poseidon::MdsOperands ops = poseidon::mds_operands();
    uint64_t s[12];
#pragma unroll
    for (int k = 0; k < 12; k++) s[k] = states[i * 12 + k];
#pragma unroll 1
    for (int l = 0; l < reps; l++) {
        if constexpr (WHICH == 0) poseidon_vector::permute(s);
        else if constexpr (WHICH == 1) poseidon::permute(s, ops);
        else poseidon::permute_blocked(s, ops);  // the blocked partial rounds (poseidon_blocked.h)
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

// ---- variant that was not kept: three v_mfma_i32_16x16x64_i8 per byte plane (four rows each; lane l = column l & 15, k-group
// l >> 4, receives rows 4 (l >> 4) .. + 3), A block-diagonal over the four k-groups -----------------------------------------
struct MdsOperands16 {
    poseidon::v4i32 A[3];
    poseidon::v4i32 C;
};
__device__ __forceinline__ MdsOperands16 mds_operands16() {
    const uint32_t lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const uint32_t m = lane & 15, g = lane >> 4, v = m & 3;
    const bool on = (m >> 2) == g;
    MdsOperands16 o;
#pragma unroll
    for (int R = 0; R < 3; R++) {
#pragma unroll
        for (int w = 0; w < 3; w++) o.A[R][w] = on ? (int)poseidon::mds_row_word((4 * w + 24 - 4 * R - v) % 12) : 0;
        o.A[R][3] = 0;
    }
    o.C = poseidon::v4i32{POSEIDON_MDS_PLANE_OFFSET, POSEIDON_MDS_PLANE_OFFSET, POSEIDON_MDS_PLANE_OFFSET, POSEIDON_MDS_PLANE_OFFSET};
    asm volatile("" : "+v"(o.A[0]), "+v"(o.A[1]), "+v"(o.A[2]), "+v"(o.C));
    return o;
}
__device__ __forceinline__ void mds_layer16(uint64_t (&s)[12], const MdsOperands16 &ops, const uint32_t *__restrict__ xy) {
    poseidon::v4i32 T[8];
#pragma unroll
    for (int G = 0; G < 3; G++)
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const uint32_t r0 = (uint32_t)(s[4 * G] >> (32 * h)), r1 = (uint32_t)(s[4 * G + 1] >> (32 * h));
            const uint32_t r2 = (uint32_t)(s[4 * G + 2] >> (32 * h)), r3 = (uint32_t)(s[4 * G + 3] >> (32 * h));
            const uint32_t a01 = __builtin_amdgcn_perm(r1, r0, 0x05010400u), c01 = __builtin_amdgcn_perm(r1, r0, 0x07030602u);
            const uint32_t a23 = __builtin_amdgcn_perm(r3, r2, 0x05010400u), c23 = __builtin_amdgcn_perm(r3, r2, 0x07030602u);
            T[4 * h + 0][G] = (int)(__builtin_amdgcn_perm(a23, a01, 0x05040100u) ^ 0x80808080u);
            T[4 * h + 1][G] = (int)(__builtin_amdgcn_perm(a23, a01, 0x07060302u) ^ 0x80808080u);
            T[4 * h + 2][G] = (int)(__builtin_amdgcn_perm(c23, c01, 0x05040100u) ^ 0x80808080u);
            T[4 * h + 3][G] = (int)(__builtin_amdgcn_perm(c23, c01, 0x07060302u) ^ 0x80808080u);
        }
    const uint32_t x0l = (uint32_t)s[0], x0h = (uint32_t)(s[0] >> 32);
#pragma unroll
    for (int R = 0; R < 3; R++) {
        poseidon::v4i32 D[8];
#pragma unroll
        for (int b = 0; b < 8; b++) D[b] = __builtin_amdgcn_mfma_i32_16x16x64_i8(ops.A[R], T[b], ops.C, 0, 0, 0);
#pragma unroll
        for (int v = 0; v < 4; v++) {
            const int r = 4 * R + v;
            const uint32_t Al = (uint32_t)D[0][v] | ((uint32_t)D[2][v] << 16), Bl = (uint32_t)D[1][v] | ((uint32_t)D[3][v] << 16);
            const uint32_t Ah = (uint32_t)D[4][v] | ((uint32_t)D[6][v] << 16), Bh = (uint32_t)D[5][v] | ((uint32_t)D[7][v] << 16);
            uint64_t al = ((uint64_t)xy[2 * r] << 32) | Al, ah = ((uint64_t)xy[2 * r + 1] << 32) | Ah;
            asm("v_mad_u64_u32 %0, vcc, %2, %4, %0\n\tv_mad_u64_u32 %1, vcc, %3, %4, %1" : "+v"(al), "+v"(ah) : "v"(Bl), "v"(Bh), "s"(256u) : "vcc");
            if (r == 0) asm("v_mad_u64_u32 %0, vcc, %2, 8, %0\n\tv_mad_u64_u32 %1, vcc, %3, 8, %1" : "+v"(al), "+v"(ah) : "v"(x0l), "v"(x0h) : "vcc");
            s[r] = gl::fold96(al, ah);
        }
    }
}
__global__ __launch_bounds__(256) void permute16_kernel(uint64_t *states, int reps) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const MdsOperands16 ops = mds_operands16();
    uint64_t s[12];
#pragma unroll
    for (int k = 0; k < 12; k++) s[k] = states[i * 12 + k];
#pragma unroll 1
    for (int l = 0; l < reps; l++) {
#pragma unroll
        for (int i2 = 0; i2 < 12; i2++) s[i2] = gl::add_canonical(s[i2], POSEIDON_ALL_ROUND_CONSTANTS[i2]);
#pragma unroll 1
        for (int r = 0; r < 30; r++) {
            if (r < 4 || r >= 26) {
#pragma unroll
                for (int i2 = 0; i2 < 12; i2++) s[i2] = gl::pow7(s[i2]);
            } else {
                s[0] = gl::pow7(s[0]);
            }
            mds_layer16(s, ops, POSEIDON_MDS_XY + 24 * r);
        }
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
    uint64_t *d[3];
    for (auto &p : d)
        if (hipMalloc(&p, n * 96) != hipSuccess) return 2;
    std::vector<uint64_t> out[3];
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
    // The clock moves with the load (the same kernel is up to 10 % faster as the fourth back-to-back launch than as the second, and
    // slower again after an idle gap): every variant runs eight launches back to back, about 50 ms like the leaf hashing of a
    // commit, and the last six are timed together.
    for (int reps : {4, 4}) {
        float t[3];
        for (int w = 0; w < 3; w++) {
            (void)hipMemcpy(d[w], h.data(), n * 96, hipMemcpyHostToDevice);
            for (int pass = 0; pass < 8; pass++) {
                if (pass == 2) (void)hipEventRecord(e[0]);
                if (w == 0) hipLaunchKernelGGL(permute_kernel<0>, dim3(n / 256), dim3(256), 0, 0, d[w], reps);
                else if (w == 1) hipLaunchKernelGGL(permute_kernel<1>, dim3(n / 256), dim3(256), 0, 0, d[w], reps);
                else hipLaunchKernelGGL(permute_kernel<2>, dim3(n / 256), dim3(256), 0, 0, d[w], reps);
            }
            (void)hipEventRecord(e[1]);
            (void)hipDeviceSynchronize();
            (void)hipEventElapsedTime(&t[w], e[0], e[1]);
            t[w] /= 6;
            (void)hipMemcpy(out[w].data(), d[w], n * 96, hipMemcpyDeviceToHost);
        }
        uint64_t bad = 0, bad16 = 0;
        for (uint64_t k = 0; k < n * 12; k++) bad += out[0][k] != out[1][k], bad16 += out[0][k] != out[2][k];
        printf("{\"permutations_per_state_and_launch\": %d, \"launches\": 8, \"states\": %llu, \"vector_ms\": %.3f, \"matrix_core_ms\": %.3f, "
               "\"matrix_core_blocked_partial_rounds_ms\": %.3f, \"G_perm_per_s\": [%.3f, %.3f, %.3f], \"mismatching_words_after_32_permutations\": [%llu, %llu]}\n",
               reps, (unsigned long long)n, t[0], t[1], t[2], n * reps / t[0] / 1e6, n * reps / t[1] / 1e6, n * reps / t[2] / 1e6, (unsigned long long)bad,
               (unsigned long long)bad16);
    }
    return 0;
}
