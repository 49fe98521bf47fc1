// ubench_issue.hip — which resource binds the field-arithmetic kernels on gfx950: issue slots or the multiplier?
// Loop bodies with a controlled mix of v_mad_u64_u32 and cheap VALU instructions (four independent chains per
// thread, 8 waves per SIMD). Reports SIMD cycles per wave and loop iteration; comparing mixes shows what an extra
// multiply-add costs and whether cheap instructions hide in its shadow.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench_issue.hip -o tools/ubench_issue
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

enum Kind { MOV, ADD, ADDC, CND, LSHLADD64, MULLO, MULHI, XOR, CND32, ADD32, SUB32, ADDE64, MAD1, DOT4, MAD24, PERM, ADD3, LSHLADD32, ALIGNBIT, BFE, PLSWAP32, MADI64, MADU16, LSHLOR, SDWA_PACK, XOR_LIT, OR_SDWA, SDWA_BYTE };

template <int KIND>
__device__ __forceinline__ void cheap(uint32_t &c, uint64_t &w, uint32_t x) {
    if constexpr (KIND == MOV) asm volatile("v_mov_b32_e32 %0, %1" : "=v"(c) : "v"(x));
    if constexpr (KIND == ADD) asm volatile("v_add_co_u32_e32 %0, vcc, %0, %1" : "+v"(c) : "v"(x) : "vcc");
    if constexpr (KIND == ADDC) asm volatile("v_addc_co_u32_e32 %0, vcc, %0, %1, vcc" : "+v"(c) : "v"(x) : "vcc");
    if constexpr (KIND == CND) asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(c) : "v"(x) : "vcc");
    if constexpr (KIND == LSHLADD64) asm volatile("v_lshl_add_u64 %0, %0, 3, %0" : "+v"(w));
    if constexpr (KIND == MULLO) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(c) : "v"(x));
    if constexpr (KIND == MULHI) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(c) : "v"(x));
    if constexpr (KIND == XOR) asm volatile("v_xor_b32_e32 %0, %0, %1" : "+v"(c) : "v"(x));
    if constexpr (KIND == CND32) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(c) : "v"(x) : "vcc");   // VOP2: mask implicitly vcc
    if constexpr (KIND == ADD32) asm volatile("v_add_u32_e32 %0, %0, %1" : "+v"(c) : "v"(x));                     // no carry-out
    if constexpr (KIND == SUB32) asm volatile("v_sub_u32_e32 %0, %0, %1" : "+v"(c) : "v"(x));
    if constexpr (KIND == ADDE64) asm volatile("v_add_co_u32_e64 %0, vcc, %0, %1" : "+v"(c) : "v"(x) : "vcc");    // VOP3 encoding of the carry add
    if constexpr (KIND == DOT4) asm volatile("v_dot4_u32_u8 %0, %1, %1, %0" : "+v"(c) : "v"(x));                  // 4 x (u8 * u8) + u32
    if constexpr (KIND == MAD24) asm volatile("v_mad_u32_u24 %0, %1, %1, %0" : "+v"(c) : "v"(x));                 // 24 x 24 + 32
    if constexpr (KIND == PERM) asm volatile("v_perm_b32 %0, %0, %1, %1" : "+v"(c) : "v"(x));                     // byte shuffle
    if constexpr (KIND == ADD3) asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(c) : "v"(x));
    if constexpr (KIND == LSHLADD32) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(c) : "v"(x));
    if constexpr (KIND == ALIGNBIT) asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(c) : "v"(x));
    if constexpr (KIND == BFE) asm volatile("v_bfe_u32 %0, %0, 3, 21" : "+v"(c));
    if constexpr (KIND == PLSWAP32) { uint32_t t = x; asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(c), "+v"(t)); }
    if constexpr (KIND == MADI64) asm volatile("v_mad_i64_i32 %0, vcc, %1, %1, %0" : "+v"(w) : "v"(x) : "vcc");
    if constexpr (KIND == MADU16) asm volatile("v_mad_u32_u16 %0, %1, %1, %0" : "+v"(c) : "v"(x));
    // round 6: the MDS layer's packing (P0 | P2 << 16) as the VOP3 instruction it uses, and as sub-dword forms of VOP1 / VOP2 (SDWA)
    if constexpr (KIND == LSHLOR) asm volatile("v_lshl_or_b32 %0, %1, 16, %0" : "+v"(c) : "v"(x));
    if constexpr (KIND == SDWA_PACK) asm volatile("v_mov_b32_sdwa %0, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0" : "+v"(c) : "v"(x));
    if constexpr (KIND == OR_SDWA) asm volatile("v_or_b32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "+v"(c) : "v"(x));
    if constexpr (KIND == SDWA_BYTE) asm volatile("v_mov_b32_sdwa %0, %1 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_1" : "+v"(c) : "v"(x));
    if constexpr (KIND == XOR_LIT) asm volatile("v_xor_b32_e32 %0, 0x80808080, %0" : "+v"(c));   // VOP2 + 32-bit literal: an 8-byte encoding
    if constexpr (KIND == MAD1) asm volatile("v_mad_u64_u32 %0, vcc, %1, 1, %0" : "+v"(w) : "v"(x) : "vcc");       // multiply-add used as a 64-bit add
}

template <int MADS, int CHEAP, int KIND>
__global__ __launch_bounds__(256) void mix_kernel(uint64_t *out, uint32_t seed, int iters) {
    uint32_t x = (threadIdx.x + blockIdx.x * 256u) * 2654435761u + seed, y = x ^ 0x9E3779B9u;
    uint64_t a[4] = {x, y, (uint64_t)x + y, (uint64_t)x * 3};
    uint32_t c[4] = {x, y, x + 1, y + 1};
    uint64_t w[4] = {x, y, x, y};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int m = 0; m < MADS; m++) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a[m & 3]) : "v"(x), "v"(y) : "vcc");
#pragma unroll
        for (int k = 0; k < CHEAP; k++) cheap<KIND>(c[k & 3], w[k & 3], x);
    }
    out[threadIdx.x + blockIdx.x * 256u] = a[0] ^ a[1] ^ a[2] ^ a[3] ^ c[0] ^ c[1] ^ c[2] ^ c[3] ^ w[0] ^ w[1] ^ w[2] ^ w[3];
}

static int g_cus, g_mhz;
static uint64_t *g_out;

template <int MADS, int CHEAP, int KIND>
void run(const char *label, int waves_per_simd = 8) {
    const int iters = 4096, blocks = g_cus * waves_per_simd;  // 4 waves per block, 4 SIMDs per CU
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int r = 0; r < 4; r++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((mix_kernel<MADS, CHEAP, KIND>), dim3(blocks), dim3(256), 0, 0, g_out, 17u + r, iters);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (r && ms < best) best = ms;
    }
    // every SIMD runs waves_per_simd waves to completion: cycles per wave-iteration = time * clock / (waves * iters)
    double cyc = best * 1e-3 * g_mhz * 1e6 / ((double)waves_per_simd * iters);
    printf("%-44s mads %2d cheap %2d  waves/SIMD %d  %8.3f ms  %7.2f cycles per wave-iteration  (%.2f per instruction)\n", label, MADS, CHEAP,
           waves_per_simd, best, cyc, cyc / (MADS + CHEAP));
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    g_cus = prop.multiProcessorCount;
    g_mhz = prop.clockRate / 1000;
    printf("device: %s, CUs %d, clock %d MHz (cycle figures assume this clock)\n", prop.name, g_cus, g_mhz);
    CK(hipMalloc(&g_out, (size_t)g_cus * 16 * 256 * 8));
    run<16, 0, MOV>("v_mad_u64_u32 only");
    run<0, 16, MOV>("v_mov_b32 only");
    run<0, 16, ADD>("v_add_co_u32 only");
    run<0, 16, ADDC>("v_addc_co_u32 only");
    run<0, 16, CND>("v_cndmask_b32 only");
    run<0, 16, XOR>("v_xor_b32 only");
    run<0, 16, LSHLADD64>("v_lshl_add_u64 only");
    run<0, 16, MULLO>("v_mul_lo_u32 only");
    run<0, 16, MULHI>("v_mul_hi_u32 only");
    run<0, 16, CND32>("v_cndmask_b32_e32 (VOP2, implicit vcc) only");
    run<0, 16, ADD32>("v_add_u32 (no carry-out) only");
    run<0, 16, SUB32>("v_sub_u32 (no carry-out) only");
    run<0, 16, ADDE64>("v_add_co_u32_e64 (VOP3) only");
    run<0, 16, MAD1>("v_mad_u64_u32 x*1 + acc only");
    run<0, 16, DOT4>("v_dot4_u32_u8 only");
    run<0, 16, MAD24>("v_mad_u32_u24 only");
    run<0, 16, MADU16>("v_mad_u32_u16 only");
    run<0, 16, PERM>("v_perm_b32 only");
    run<0, 16, ADD3>("v_add3_u32 only");
    run<0, 16, LSHLADD32>("v_lshl_add_u32 only");
    run<0, 16, ALIGNBIT>("v_alignbit_b32 only");
    run<0, 16, BFE>("v_bfe_u32 only");
    run<0, 16, PLSWAP32>("v_permlane32_swap_b32 only");
    run<0, 16, MADI64>("v_mad_i64_i32 only");
    run<0, 16, LSHLOR>("v_lshl_or_b32 only");
    run<0, 16, SDWA_PACK>("v_mov_b32_sdwa word -> word (pack) only");
    run<0, 16, OR_SDWA>("v_or_b32_sdwa only");
    run<0, 16, SDWA_BYTE>("v_mov_b32_sdwa byte -> byte only");
    run<0, 16, XOR_LIT>("v_xor_b32 with a 32-bit literal only");
    run<0, 16, LSHLOR>("v_lshl_or_b32 only", 4);
    run<0, 16, SDWA_PACK>("v_mov_b32_sdwa word -> word (pack) only", 4);
    run<0, 16, XOR>("v_xor_b32 only", 4);
    run<0, 16, PERM>("v_perm_b32 only", 4);
    printf("-- the field multiplication's shape: 5 multiply-adds and 11-14 carry-chain instructions\n");
    run<5, 14, ADDC>("old gl::mul shape (5 + 14)");
    run<5, 11, ADDC>("new gl::mul shape (5 + 11)");
    run<5, 0, ADDC>("its multiply-adds alone");
    run<0, 11, ADDC>("its carry chain alone");
    run<4, 13, ADDC>("one multiply-add traded for 2 cheap (4 + 13)");
    run<4, 15, ADDC>("one multiply-add traded for 4 cheap (4 + 15)");
    run<5, 14, MOV>("5 + 14 v_mov");
    run<5, 11, MOV>("5 + 11 v_mov");
    printf("-- occupancy\n");
    run<5, 11, ADDC>("new gl::mul shape", 4);
    run<5, 11, ADDC>("new gl::mul shape", 2);
    run<5, 11, ADDC>("new gl::mul shape", 1);
    run<16, 0, MOV>("v_mad_u64_u32 only", 1);
    run<0, 16, ADDC>("v_addc_co_u32 only", 1);
    return 0;
}
