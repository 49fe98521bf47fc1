// mds_natural.h — EXPERIMENT of round 6 (measured, not adopted; LABNOTES 13): the MDS layer with the state fed to the matrix cores
// as it lies in the registers, included by csrc/poseidon.h in place of its own mds_layer when the build defines
// POSEIDON_MDS_NATURAL="../../tools/experiments/mds_natural.h" (tools/gpu_runs/build_variant_files.sh natural ... merkle fri).
// Bit-exact (tests/test_gpu_merkle.py: 70 passed on the variant), 194 vector instructions per layer instead of 242, and SLOWER:
// 2.18-2.25 against 2.32-2.36 G permutations/s, configs[2] commit 69.6-69.9 against 64.5 ms on one device in one run
// (profiles/r06_poseidon_natural_layout_ab.jsonl). Eighteen matrix instructions per layer instead of eight: at this density their
// 32 cycles of pipe time each are no longer hidden behind the other waves' vector instructions (about 17 cycles each show).
// ---- round 6: the state goes into the matrix cores AS IT LIES -------------------------------------------------------------
// Up to round 5 a layer first transposed the state into byte planes (one dword = the same byte of four words: 48 v_perm_b32
// per layer) so that ONE product per plane served all twelve words. The transposition is vector-ALU work and the vector ALU
// is what the hashing is bound by; the matrix pipe ran at 14 %. Now a B operand is four state dwords as they are — the low
// (or high) halves of words 4G..4G+3, sixteen bytes (j', t) = byte t of word 4G + j' — and the selection of the byte position
// moves into the A operand: result row (r', b) of row group R is  sum_{j', t} [t == b] CIRC[(4G + j' - 4R - r') mod 12] byte,
// three products (G = 0, 1, 2) chained through the accumulator per (R, half): eighteen matrix instructions per layer instead
// of eight, each with a matrix three quarters zeros — and the matrix depends on (G - R) mod 3 only (the MDS matrix is circulant),
// three A operands in all. Vector ALU per layer: 24 x (^ 0x80) + 96 packing + 72 fold96 + 2 = 194 instead of 242.
struct MdsOperands {
    v4i32 A[3];  // A[d]: this lane's row of the operand for chunks with (G - R) mod 3 == d, in its own half's sixteen k, or zero
    v16i32 C;    // 128 * (row sum) in every element
};

__device__ __forceinline__ uint32_t mds_circ(uint32_t o) {  // CIRC[o], o < 12, as a chain of selects on literals
    constexpr uint32_t WORDS[12] = POSEIDON_MDS_ROW_WORDS;  // word o = bytes CIRC[o], CIRC[o+1], ..
    uint32_t x = 0;
#pragma unroll
    for (int k = 0; k < 12; k++) x = o == (uint32_t)k ? (WORDS[k] & 0xFFu) : x;
    return x;
}

// Pure function of the lane number; call it before anything diverges.
__device__ __forceinline__ MdsOperands mds_operands() {
    const uint32_t lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const uint32_t m = lane & 31, h = lane >> 5;
    // Result register q of lane (n, h) is row (q & 3) + 8 (q >> 2) + 4 h of the product; q = 4 r' + b. Row m of A therefore
    // belongs to q = (m & 3) + 4 (m >> 3), read by the lanes of half (m >> 2) & 1, and is non-zero in that half's k only.
    const uint32_t q = (m & 3) + 4 * (m >> 3), rp = q >> 2, b = q & 3;
    const bool on = ((m >> 2) & 1) == h;
    MdsOperands o;
#pragma unroll
    for (int d = 0; d < 3; d++)
#pragma unroll
        for (int w = 0; w < 4; w++) o.A[d][w] = on ? (int)(mds_circ((4 * d + w + 12 - rp) % 12) << (8 * b)) : 0;  // dword w = word j' = w of the chunk
#pragma unroll
    for (int k = 0; k < 16; k++) o.C[k] = POSEIDON_MDS_PLANE_OFFSET;
    asm volatile("" : "+v"(o.A[0]), "+v"(o.A[1]), "+v"(o.A[2]), "+v"(o.C));
    return o;
}

__device__ __forceinline__ void require_full_wave() {
    if (__builtin_amdgcn_read_exec() != ~0ull) __builtin_trap();
}

// MDS layer + the additive constants of whatever follows, xy = [12][X, Y] (poseidon_limb_constants.h).
__device__ __forceinline__ void mds_layer(uint64_t (&s)[W], const MdsOperands &ops, const uint32_t *__restrict__ xy) {
    v4i32 B[2][3];  // B[half][G] = half `half` of words 4G .. 4G+3, every byte - 128 (the matrix cores read SIGNED bytes)
#pragma unroll
    for (int G = 0; G < 3; G++)
#pragma unroll
        for (int j = 0; j < 4; j++) {
            B[0][G][j] = (int)((uint32_t)s[4 * G + j] ^ 0x80808080u);
            B[1][G][j] = (int)((uint32_t)(s[4 * G + j] >> 32) ^ 0x80808080u);
        }
    const uint32_t x0l = (uint32_t)s[0], x0h = (uint32_t)(s[0] >> 32);
    uint64_t a[2][W];  // a[0] = sum of planes 0-3 weighted 2^(8b), a[1] = planes 4-7: the two 64-bit columns of fold96
#pragma unroll
    for (int R = 0; R < 3; R++)
#pragma unroll
        for (int half = 0; half < 2; half++) {
            v16i32 D = __builtin_amdgcn_mfma_i32_32x32x32_i8(ops.A[(3 - R) % 3], B[half][0], ops.C, 0, 0, 0);
            D = __builtin_amdgcn_mfma_i32_32x32x32_i8(ops.A[(4 - R) % 3], B[half][1], D, 0, 0, 0);
            D = __builtin_amdgcn_mfma_i32_32x32x32_i8(ops.A[(5 - R) % 3], B[half][2], D, 0, 0, 0);
#pragma unroll
            for (int rp = 0; rp < 4; rp++) {
                const int r = 4 * R + rp;
                const uint32_t even = (uint32_t)D[4 * rp] | ((uint32_t)D[4 * rp + 2] << 16), odd = (uint32_t)D[4 * rp + 1] | ((uint32_t)D[4 * rp + 3] << 16);
                a[half][r] = ((uint64_t)xy[2 * r + half] << 32) | even;
                asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a[half][r]) : "v"(odd), "s"(256u) : "vcc");
            }
        }
    // the diagonal entry stays out of the matrix product (it would push a plane's sum past 16 bits)
    asm("v_mad_u64_u32 %0, vcc, %2, %4, %0\n\t"
        "v_mad_u64_u32 %1, vcc, %3, %4, %1"
        : "+v"(a[0][0]), "+v"(a[1][0])
        : "v"(x0l), "v"(x0h), "n"(POSEIDON_MDS_DIAG0)
        : "vcc");
#pragma unroll
    for (int r = 0; r < W; r++) s[r] = gl::fold96(a[0][r], a[1][r]);  // a0 + a1 2^32 mod p; a0 < 2^41 + X 2^32, a1 < 2^41 + Y 2^32: X, Y leave the room
}
