// Experiment (round 4), NOT part of the product: the column pass's twiddle w_16G^(kB w) in front of its last radix 16 as compile-time
// shifts at the READER of the cross-wave exchange (register = w, kB = kAB >> 4 wave-uniform: a scalar switch over kB), signs absorbed
// by a radix 16 that tracks a sign per slot. Correct (tests/test_gpu_ntt.py, golden files) and 65 vector instructions per lane and
// tile fewer, but SLOWER: behind the switch's merge the compiler spills the eight kept inter-pass twiddles to scratch, and a scratch
// reload waits for the prefetched tile as well (2^20 natural 0.569 ms against 0.546; LABNOTES 11). To try it again: include this
// file in ntt_kernels.h and call shift_twiddles_radix16<39 * 4 / G, kB>(B) in place of radix_dif<4, 0>(B) (ntt_direct.hip, column
// pass, G = 2 / 4, no cosets), dropping the writer's t2 multiplication.
// ---- radix 16 whose inputs carry SIGNS (column pass: the twiddles in front of the last radix 16 as shifts) -------------------------
// Bit i of SG: slot i holds MINUS its value. A butterfly (a, c) with signs (sa, sc), tau = sa sc, and a stage twiddle of sign nu:
//   slot i0 <- a + tau c          (the difference a - c when tau = -1), sign sa
//   slot i1 <- |nu| (a - tau c) 2^K: tau = +1: a - c, or c - a when nu = -1 (sign sa); tau = -1: a + c, sign nu sa
// so no instruction is spent on a sign; slot 0 never changes its sign and every other slot ends as the i1 of a butterfly whose i0 is
// positive: after the last stage (no twiddles) all signs are +, whatever came in — as long as slot 0 came in positive.
constexpr unsigned radix16_signs_after(unsigned sg, int s) {
    const int half = 1 << s;
    for (int bb = 0; bb < 8; bb++) {
        const int i0 = (bb / half) * 2 * half + (bb % half), i1 = i0 + half;
        const bool sa = (sg >> i0) & 1, sc = (sg >> i1) & 1, nu = (39 * (bb % half) * (32 >> s)) % 192 >= 96;
        const bool s1 = (sa != sc && nu) ? !sa : sa;
        sg = (sg & ~(1u << i1)) | ((unsigned)s1 << i1);
    }
    return sg;
}
template <unsigned SG, int s>
__device__ __forceinline__ void radix16_stage_signed(uint64_t (&v)[16]) {
    constexpr int half = 1 << s;
    constexpr auto I0 = [](int bb) { return (bb / half) * 2 * half + (bb % half); };
    constexpr auto KOF = [](int bb) { return (39 * (bb % half) * (32 >> s)) % 192; };
    constexpr auto TAU = [](int bb) { return (((SG >> ((bb / half) * 2 * half + (bb % half))) ^ (SG >> ((bb / half) * 2 * half + (bb % half) + half))) & 1u) != 0; };
    gl::rare_mask f0[8], f1[8], fm[8];   // f0 / f1: masks of the values written to slots i0 / i1; fm: of the shift
    static_for<0, 8>([&](auto B_) {
        constexpr int bb = decltype(B_)::value, i0 = I0(bb), i1 = i0 + half;
        if constexpr (TAU(bb)) gl::bfly_f<false>(v[i0], v[i1], v[i1], v[i0], f1[bb], f0[bb]);        // i0 <- a - c, i1 <- a + c
        else gl::bfly_f<(KOF(bb) >= 96)>(v[i0], v[i1], v[i0], v[i1], f0[bb], f1[bb]);                // i0 <- a + c, i1 <- +-(a - c)
    });
    static_for<0, 8>([&](auto B_) {
        constexpr int bb = decltype(B_)::value, i1 = I0(bb) + half, KK = KOF(bb) % 96;
        v[i1] = gl::mul_pow2_f<KK>(v[i1], fm[bb]);   // KK = 0: nothing, mask 0
    });
    __builtin_amdgcn_sched_barrier(RARE_FENCE_MASK);
    gl::rare_mask any = 0;
    static_for<0, 8>([&](auto B_) { constexpr int bb = decltype(B_)::value; any |= f0[bb] | f1[bb] | fm[bb]; });
    if (GL_RARE_ANY(any))
        static_for<0, 8>([&](auto B_) {
            constexpr int bb = decltype(B_)::value, i0 = I0(bb), i1 = i0 + half, KK = KOF(bb) % 96;
            v[i0] = TAU(bb) ? gl::sub_fix(v[i0], f0[bb]) : gl::add_fix(v[i0], f0[bb]);
            v[i1] = gl::mul_pow2_fix<KK>(v[i1], fm[bb]);
            const uint64_t c = gl::masked_const<gl::eps_times_pow2(KK)>(f1[bb]);   // the pending e of the sum / difference, through the shift
            v[i1] = TAU(bb) ? gl::add(v[i1], c) : gl::sub(v[i1], c);
        });
}
// v[w] *= 2^(UNIT KB w) (w = 1 .. 15: the twiddle w_16G^(KB w) of the column pass, UNIT = 39 * 4 / G), then the radix 16 over w
template <int UNIT, int KB>
__device__ __forceinline__ void shift_twiddles_radix16(uint64_t (&v)[16]) {
    constexpr auto KW = [](int w) { return (UNIT * KB * w) % 192; };
    constexpr auto signs = [] { unsigned sg = 0; for (int w = 1; w < 16; w++) sg |= (unsigned)((UNIT * KB * w) % 192 >= 96) << w; return sg; };
    {
        gl::rare_mask fm[15];
        static_for<1, 16>([&](auto W_) {
            constexpr int w = decltype(W_)::value, KK = KW(w) % 96;
            v[w] = gl::mul_pow2_f<KK>(v[w], fm[w - 1]);
        });
        __builtin_amdgcn_sched_barrier(RARE_FENCE_MASK);
        gl::rare_mask any = 0;
        static_for<0, 15>([&](auto I_) { any |= fm[decltype(I_)::value]; });
        if (GL_RARE_ANY(any))
            static_for<1, 16>([&](auto W_) {
                constexpr int w = decltype(W_)::value, KK = KW(w) % 96;
                v[w] = gl::mul_pow2_fix<KK>(v[w], fm[w - 1]);
            });
    }
    constexpr unsigned SG3 = signs(), SG2 = radix16_signs_after(SG3, 3), SG1 = radix16_signs_after(SG2, 2), SG0 = radix16_signs_after(SG1, 1);
    static_assert(radix16_signs_after(SG0, 0) == 0, "every sign is absorbed");
    radix16_stage_signed<SG3, 3>(v);
    radix16_stage_signed<SG2, 2>(v);
    radix16_stage_signed<SG1, 1>(v);
    radix16_stage_signed<SG0, 0>(v);
}

