// mds_interleave.h — EXPERIMENT of round 6 (measured, not adopted; LABNOTES 13.B): csrc/poseidon.h's mds_layer with its eight matrix
// instructions issued in PAIRS between pieces of vector work that do not depend on them, pinned with sched_barrier. Included in place of
// the product's mds_layer when the build defines POSEIDON_MDS_LAYER="../../tools/experiments/mds_interleave.h"
// (tools/gpu_runs/build_variant_files.sh interleave ... merkle fri). Bit-exact (tests/test_gpu_merkle.py on the variant) and no faster:
// 2.33-2.35 against 2.36-2.39 G permutations/s in permute_batch_kernel (128 registers, four waves per SIMD, no spills); the hashing
// kernels need 138-140 registers with it (three waves per SIMD: commit 65.6 against 62.3 ms). profiles/r06_poseidon_interleave_ab.jsonl
// EXPERIMENT (round 6): the same layer with its matrix instructions issued in PAIRS between pieces of vector work that do not depend
// on them — planes (0, 2) | transposition of the high halves | planes (4, 6) | pack the even planes of the low half | planes (1, 3) |
// pack the even planes of the high half | planes (5, 7) | the odd planes, multiply-adds, fold — so that a wave waits for a matrix
// result at one place instead of two and never with four instructions queued. Four result tiles live at any time, as before.
__device__ __forceinline__ void mds_layer(uint64_t (&s)[W], const MdsOperands &ops, const uint32_t *__restrict__ xy) {
    v4i32 T[8];
    auto transpose_half = [&](auto H_) {
        constexpr int h = decltype(H_)::value;
#pragma unroll
        for (int G = 0; G < 3; G++) {
            const uint32_t r0 = (uint32_t)(s[4 * G] >> (32 * h)), r1 = (uint32_t)(s[4 * G + 1] >> (32 * h));
            const uint32_t r2 = (uint32_t)(s[4 * G + 2] >> (32 * h)), r3 = (uint32_t)(s[4 * G + 3] >> (32 * h));
            const uint32_t a01 = __builtin_amdgcn_perm(r1, r0, 0x05010400u), c01 = __builtin_amdgcn_perm(r1, r0, 0x07030602u);
            const uint32_t a23 = __builtin_amdgcn_perm(r3, r2, 0x05010400u), c23 = __builtin_amdgcn_perm(r3, r2, 0x07030602u);
            T[4 * h + 0][G] = (int)(__builtin_amdgcn_perm(a23, a01, 0x05040100u) ^ 0x80808080u);
            T[4 * h + 1][G] = (int)(__builtin_amdgcn_perm(a23, a01, 0x07060302u) ^ 0x80808080u);
            T[4 * h + 2][G] = (int)(__builtin_amdgcn_perm(c23, c01, 0x05040100u) ^ 0x80808080u);
            T[4 * h + 3][G] = (int)(__builtin_amdgcn_perm(c23, c01, 0x07060302u) ^ 0x80808080u);
        }
    };
    const uint32_t x0l = (uint32_t)s[0], x0h = (uint32_t)(s[0] >> 32);
    uint32_t ev[2][W];
    uint64_t al[W], ah[W];
    transpose_half(std::integral_constant<int, 0>{});
    v16i32 Da = __builtin_amdgcn_mfma_i32_32x32x32_i8(ops.A, T[0], ops.C, 0, 0, 0), Db = __builtin_amdgcn_mfma_i32_32x32x32_i8(ops.A, T[2], ops.C, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    transpose_half(std::integral_constant<int, 1>{});
    v16i32 Dc = __builtin_amdgcn_mfma_i32_32x32x32_i8(ops.A, T[4], ops.C, 0, 0, 0), Dd = __builtin_amdgcn_mfma_i32_32x32x32_i8(ops.A, T[6], ops.C, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < W; r++) ev[0][r] = (uint32_t)Da[r] | ((uint32_t)Db[r] << 16);
    __builtin_amdgcn_sched_barrier(0);
    Da = __builtin_amdgcn_mfma_i32_32x32x32_i8(ops.A, T[1], ops.C, 0, 0, 0), Db = __builtin_amdgcn_mfma_i32_32x32x32_i8(ops.A, T[3], ops.C, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < W; r++) ev[1][r] = (uint32_t)Dc[r] | ((uint32_t)Dd[r] << 16);
    __builtin_amdgcn_sched_barrier(0);
    Dc = __builtin_amdgcn_mfma_i32_32x32x32_i8(ops.A, T[5], ops.C, 0, 0, 0), Dd = __builtin_amdgcn_mfma_i32_32x32x32_i8(ops.A, T[7], ops.C, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < W; r++) {
        const uint32_t Bl = (uint32_t)Da[r] | ((uint32_t)Db[r] << 16);
        al[r] = ((uint64_t)xy[2 * r] << 32) | ev[0][r];
        asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(al[r]) : "v"(Bl), "s"(256u) : "vcc");
    }
#pragma unroll
    for (int r = 0; r < W; r++) {
        const uint32_t Bh = (uint32_t)Dc[r] | ((uint32_t)Dd[r] << 16);
        ah[r] = ((uint64_t)xy[2 * r + 1] << 32) | ev[1][r];
        asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(ah[r]) : "v"(Bh), "s"(256u) : "vcc");
    }
    asm("v_mad_u64_u32 %0, vcc, %2, %4, %0\n\t"
        "v_mad_u64_u32 %1, vcc, %3, %4, %1"
        : "+v"(al[0]), "+v"(ah[0])
        : "v"(x0l), "v"(x0h), "n"(POSEIDON_MDS_DIAG0)
        : "vcc");
#pragma unroll
    for (int r = 0; r < W; r++) s[r] = gl::fold96(al[r], ah[r]);
}
