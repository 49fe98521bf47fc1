// gate_emit.hip — emitters of the gate register programs behind the C ABI (host code only).
//
// The device evaluates evaluate_gate_constraints_base_batch (plonky2/src/plonk/vanishing_poly.rs:267-306) for any circuit from
// one register program per gate (include/plonky2_hip.h, GlGateInstr). Until round 3 the only producer of such programs was
// the Python host (plonky2_gpu_amd/gate_program.py): a Rust or C++ host could prove exactly one circuit, the compiled-in
// ed25519 table. gl_gate_programs_emit builds the programs of a gate list natively, for the twelve gate kinds of that list and
// (round 5) the eight other gates of upstream plonky2 — what a Rust host would otherwise have to port from each gate's
// eval_unfiltered_base_one:
//   Noop, Constant, PublicInput, Arithmetic        plonky2/src/gates/{noop,constant,public_input,arithmetic_base}.rs
//   BaseSum<B>, RandomAccess, Poseidon             plonky2/src/gates/{base_sum,random_access,poseidon}.rs
//   U32AddMany, U32Arithmetic, U32Subtraction, U32RangeCheck, Comparison        u32/src/gates/*.rs
//   ArithmeticExtension, MulExtension, Reducing, ReducingExtension, Exponentiation, PoseidonMds, Low/HighDegreeInterpolation
//                                                  plonky2/src/gates/{arithmetic_extension,multiplication_extension,reducing,
//                                                  reducing_extension,exponentiation,poseidon_mds,low_degree_interpolation,high_degree_interpolation}.rs
// The output is identical, instruction for instruction and immediate for immediate, to gate_program.py's
// (tests/test_gate_emit.py), which the oracle's gate restatements check (tests/test_gate_programs_cpu.py).
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/plonky2_hip.h"
#define POSEIDON_CONST static const
#include "poseidon_constants.h"

namespace {

typedef unsigned __int128 u128;
constexpr uint64_t P = 0xFFFFFFFF00000001ull;
enum : uint16_t { LOAD_WIRE = 0, LOAD_CONST, LOAD_PI, LOAD_IMM, ADD, SUB, MUL, EMIT, MULK, ACC, ACCR };
constexpr int MAX_REGS = 64;
const u128 ACC_LIMIT = (u128)1 << 63;  // each half of an accumulator stays below this (gl::fold96's precondition)

// Field constants referenced by LOAD_IMM / ACC, shared by all gates of a circuit (deduplicated)
struct ImmediatePool {
    std::vector<uint64_t> values;
    std::map<uint64_t, uint32_t> index_of;
    uint32_t index(u128 value) {
        const uint64_t v = (uint64_t)(value % P);
        auto it = index_of.find(v);
        if (it != index_of.end()) return it->second;
        if (values.size() >= 65536) throw std::runtime_error("more than 65536 distinct immediates");
        const uint32_t i = (uint32_t)values.size();
        index_of[v] = i;
        values.push_back(v);
        return i;
    }
};

u128 ipow(u128 base, unsigned e) {
    u128 r = 1;
    for (unsigned i = 0; i < e; i++) r *= base;
    return r;
}

// Emits one gate's register program. Registers come from a free list (lowest first; freed registers are reused last in,
// first out); release() returns all of them (between independent constraints).
struct GateAsm {
    std::vector<GlGateInstr> instrs;
    ImmediatePool *pool;
    std::vector<int> free_list;
    u128 acc_bound[4] = {0, 0, 0, 0};

    explicit GateAsm(ImmediatePool *p = nullptr) : pool(p) { release(); }
    void release() {
        free_list.clear();
        for (int r = MAX_REGS - 1; r >= 0; r--) free_list.push_back(r);
    }
    int reg() {
        if (free_list.empty()) throw std::runtime_error("gate program needs more than 64 live registers");
        const int r = free_list.back();
        free_list.pop_back();
        return r;
    }
    void free1(int r) {
        if (std::find(free_list.begin(), free_list.end(), r) != free_list.end()) throw std::runtime_error("register freed twice");
        free_list.push_back(r);
    }
    void free(std::initializer_list<int> regs) {
        for (int r : regs) free1(r);
    }
    // operands are 16-bit fields of the instruction word: an index that does not fit is an error, never a wrapped "valid" program
    static uint16_t field(long v) {
        if (v < 0 || v > 0xFFFF) throw std::runtime_error("gate program operand (wire, constant, immediate or register index) does not fit 16 bits");
        return (uint16_t)v;
    }
    int op(uint16_t code, int a, int b = 0, int dst = -1) {
        const int r = dst < 0 ? reg() : dst;
        instrs.push_back(GlGateInstr{code, field(r), field(a), field(b)});
        return r;
    }
    int wire(int i) { return op(LOAD_WIRE, i); }
    int constant(int i) { return op(LOAD_CONST, i); }
    int pi(int i) { return op(LOAD_PI, i); }
    ImmediatePool &need_pool() {
        if (!pool) throw std::runtime_error("this gate needs an ImmediatePool");
        return *pool;
    }
    int imm(u128 value) { return op(LOAD_IMM, (int)need_pool().index(value)); }
    int add(int a, int b, int dst = -1) { return op(ADD, a, b, dst); }
    int sub(int a, int b, int dst = -1) { return op(SUB, a, b, dst); }
    int mul(int a, int b, int dst = -1) { return op(MUL, a, b, dst); }
    int mulk(int a, int shift, int dst = -1) { return op(MULK, a, shift, dst); }
    void emit(int a) { instrs.push_back(GlGateInstr{EMIT, 0, field(a), 0}); }

    // may r * weight still be added to accumulator q without either half being able to reach 2^63?
    bool acc_fits(u128 weight, int q = 0) const { return weight < ((u128)1 << 32) && acc_bound[q] + weight * 0xFFFFFFFFull < ACC_LIMIT; }
    // acc[q] += r[a] * weight, without a modular step (ACC): weight < 2^32, bound checked here
    void acc(int a, u128 weight = 1, int q = 0) {
        ImmediatePool &pl = need_pool();
        if (!acc_fits(weight, q)) throw std::runtime_error("accumulator could overflow: reduce (accr) earlier");
        acc_bound[q] += weight * 0xFFFFFFFFull;
        instrs.push_back(GlGateInstr{ACC, field(q), field(a), field((long)pl.index(weight))});
    }
    // register <- acc[q] mod p; acc[q] <- 0 (ACCR)
    int accr(int q = 0, int dst = -1) {
        acc_bound[q] = 0;
        return op(ACCR, q, 0, dst);
    }
    // sum_i r[reg_i] * weight_i through accumulator q; reduces in between only if the static bound demands it
    int weighted_sum(const std::vector<std::pair<int, uint64_t>> &terms, int q = 0, int dst = -1) {
        bool started = false;
        for (auto &t : terms) {
            if (started && !acc_fits(t.second, q)) {
                const int part = accr(q);
                acc(part, 1, q);
                free1(part);
            }
            acc(t.first, t.second, q);
            started = true;
        }
        if (!started) return op(LOAD_IMM, (int)need_pool().index(0), 0, dst);
        return accr(q, dst);
    }
    // sum terms[i] * base^i (plonk_common.rs:116-128) for an integer base; returns a fresh register. `terms` are registers, or
    // with wires = true wire indices that are loaded for the purpose. Blocks of consecutive terms whose weights stay below 2^32
    // and provably fit one accumulation, joined by Horner steps with base^(block length).
    int reduce_with_powers(const std::vector<int> &terms, uint64_t base, bool wires = false, int q = 0) {
        if (terms.empty()) return imm(0);
        std::vector<std::pair<int, unsigned>> blocks;
        size_t i = 0;
        while (i < terms.size()) {
            unsigned j = 0;
            u128 bound = 0;
            while (i + j < terms.size() && ipow(base, j) < ((u128)1 << 32) && bound + ipow(base, j) * 0xFFFFFFFFull < ACC_LIMIT) {
                bound += ipow(base, j) * 0xFFFFFFFFull;
                j++;
            }
            for (unsigned k = 0; k < j; k++) {
                const int t = wires ? wire(terms[i + k]) : terms[i + k];
                acc(t, ipow(base, k), q);
                if (wires) free1(t);
            }
            blocks.push_back({accr(q), j});
            i += j;
        }
        const int accu = blocks.back().first;
        for (size_t b = blocks.size() - 1; b-- > 0;) {
            const u128 step = ipow(base, blocks[b].second);
            unsigned bits = 0;
            while (((u128)1 << bits) < step) bits++;
            if ((step & (step - 1)) == 0 && bits < 64) {
                mulk(accu, (int)bits, accu);
            } else {
                const int m = imm(step % P);
                mul(accu, m, accu);
                free1(m);
            }
            add(accu, blocks[b].first, accu);
            free1(blocks[b].first);
        }
        return accu;
    }
    // prod_{k < len(small)} (x - k); small[k] = register holding the constant k. For four factors y (y + 2) with y = x (x - 3).
    int range_product(int x, const std::vector<int> &small) {
        if (small.size() == 4) {
            const int t = sub(x, small[3]);
            const int y = mul(x, t);
            add(y, small[2], t);
            mul(y, t, y);
            free1(t);
            return y;
        }
        const int a = sub(x, small[0]);
        for (size_t k = 1; k < small.size(); k++) {
            const int t = sub(x, small[k]);
            mul(a, t, a);
            free1(t);
        }
        return a;
    }
};

typedef std::vector<GlGateInstr> Program;

// ArithmeticGate { num_ops } (plonky2/src/gates/arithmetic_base.rs:199-216)
Program arithmetic_gate(unsigned num_ops) {
    GateAsm g;
    for (unsigned i = 0; i < num_ops; i++) {
        g.release();
        const int c0 = g.constant(0), c1 = g.constant(1);
        const int m0 = g.wire(4 * i), m1 = g.wire(4 * i + 1), ad = g.wire(4 * i + 2), out = g.wire(4 * i + 3);
        const int p0 = g.mul(m0, m1);
        const int p1 = g.mul(p0, c0);
        const int p2 = g.mul(ad, c1);
        const int computed = g.add(p1, p2);
        g.emit(g.sub(out, computed));
    }
    return g.instrs;
}

// ConstantGate { num_consts } (plonky2/src/gates/constant.rs:150-158)
Program constant_gate(unsigned num_consts) {
    GateAsm g;
    for (unsigned i = 0; i < num_consts; i++) {
        g.release();
        const int c = g.constant(i), w = g.wire(i);
        g.emit(g.sub(c, w));
    }
    return g.instrs;
}

// PublicInputGate (plonky2/src/gates/public_input.rs:129-139)
Program public_input_gate() {
    GateAsm g;
    for (int i = 0; i < 4; i++) {
        g.release();
        const int w = g.wire(i), h = g.pi(i);
        g.emit(g.sub(w, h));
    }
    return g.instrs;
}

// BaseSumGate<B> { num_limbs } (plonky2/src/gates/base_sum.rs:213-230)
Program base_sum_gate(unsigned B, unsigned num_limbs, ImmediatePool &pool) {
    GateAsm g(&pool);
    std::vector<int> limbs;
    for (unsigned i = 0; i < num_limbs; i++) limbs.push_back(1 + i);
    const int accu = g.reduce_with_powers(limbs, B, true);
    const int s = g.wire(0);
    g.emit(g.sub(accu, s));
    g.release();
    std::vector<int> small;
    for (unsigned k = 0; k < B; k++) small.push_back(g.imm(k));
    for (unsigned i = 0; i < num_limbs; i++) {
        const int x = g.wire(1 + i);
        const int p = g.range_product(x, small);
        g.emit(p);
        g.free({x, p});
    }
    return g.instrs;
}

// range-check `count` base-4 limbs (one constraint each, from the LAST limb down like the reference's `for j in (0..n).rev()`),
// and return (low, high) = the limbs below / from `split` recombined in base 4
std::pair<int, int> u32_limb_checks(GateAsm &g, unsigned first_limb_wire, unsigned count, unsigned split, const std::vector<int> &small) {
    for (unsigned j = count; j-- > 0;) {
        const int limb = g.wire(first_limb_wire + j);
        const int p = g.range_product(limb, small);
        g.emit(p);
        if (j < split)
            g.acc(limb, ipow(4, j), 0);
        else
            g.acc(limb, ipow(4, j - split), 1);
        g.free({limb, p});
    }
    const int low = split > 0 ? g.accr(0) : g.imm(0);
    const int high = count > split ? g.accr(1) : g.imm(0);
    return {low, high};
}

// U32AddManyGate (u32/src/gates/add_many_u32.rs:143-184)
Program u32_add_many_gate(unsigned num_addends, unsigned num_ops, ImmediatePool &pool) {
    GateAsm g(&pool);
    for (unsigned i = 0; i < num_ops; i++) {
        g.release();
        const unsigned o = (num_addends + 3) * i;
        std::vector<int> small;
        for (int k = 0; k < 4; k++) small.push_back(g.imm(k));
        int computed;
        if (num_addends >= 3) {  // addends + carry-in: plain sums, one fold
            for (unsigned j = 0; j < num_addends + 1; j++) {
                const int t = g.wire(o + j);
                g.acc(t, 1, 2);
                g.free1(t);
            }
            computed = g.accr(2);
        } else {
            computed = g.wire(o + num_addends);
            for (unsigned j = 0; j < num_addends; j++) {
                const int t = g.wire(o + j);
                g.add(computed, t, computed);
                g.free1(t);
            }
        }
        const int res = g.wire(o + num_addends + 1), car = g.wire(o + num_addends + 2);
        const int comb = g.mulk(car, 32);
        g.add(comb, res, comb);
        g.emit(g.sub(comb, computed, comb));
        g.free({comb, computed});
        const auto lh = u32_limb_checks(g, (num_addends + 3) * num_ops + 18 * i, 18, 16, small);
        g.emit(g.sub(lh.first, res, lh.first));
        g.emit(g.sub(lh.second, car, lh.second));
    }
    return g.instrs;
}

// U32ArithmeticGate (u32/src/gates/arithmetic_u32.rs:326-385)
Program u32_arithmetic_gate(unsigned num_ops, ImmediatePool &pool) {
    GateAsm g(&pool);
    for (unsigned i = 0; i < num_ops; i++) {
        g.release();
        std::vector<int> small;
        for (int k = 0; k < 4; k++) small.push_back(g.imm(k));
        const int one = small[1], umax = g.imm(0xFFFFFFFFull);
        const int m0 = g.wire(6 * i), m1 = g.wire(6 * i + 1), ad = g.wire(6 * i + 2), lo = g.wire(6 * i + 3), hi = g.wire(6 * i + 4),
                  inv = g.wire(6 * i + 5);
        const int computed = g.mul(m0, m1);
        g.add(computed, ad, computed);
        const int t = g.sub(umax, hi);
        g.mul(inv, t, t);
        g.sub(t, one, t);
        g.emit(g.mul(t, lo, t));
        g.mulk(hi, 32, t);
        g.add(t, lo, t);
        g.emit(g.sub(t, computed, t));
        g.free({t, computed, m0, m1, ad, inv});
        const auto lh = u32_limb_checks(g, 6 * num_ops + 32 * i, 32, 16, small);
        g.emit(g.sub(lh.first, lo, lh.first));
        g.emit(g.sub(lh.second, hi, lh.second));
    }
    return g.instrs;
}

// U32SubtractionGate (u32/src/gates/subtraction_u32.rs:233-269)
Program u32_subtraction_gate(unsigned num_ops, ImmediatePool &pool) {
    GateAsm g(&pool);
    for (unsigned i = 0; i < num_ops; i++) {
        g.release();
        std::vector<int> small;
        for (int k = 0; k < 4; k++) small.push_back(g.imm(k));
        const int one = small[1];
        const int x = g.wire(5 * i), y = g.wire(5 * i + 1), bi = g.wire(5 * i + 2), res = g.wire(5 * i + 3), bo = g.wire(5 * i + 4);
        int t = g.sub(x, y);
        g.sub(t, bi, t);
        const int u = g.mulk(bo, 32);
        g.add(t, u, t);
        g.emit(g.sub(res, t, t));
        g.free({t, u, x, y, bi});
        const auto lh = u32_limb_checks(g, 5 * num_ops + 16 * i, 16, 16, small);
        g.emit(g.sub(lh.first, res, lh.first));
        t = g.sub(one, bo);
        g.emit(g.mul(bo, t, t));
    }
    return g.instrs;
}

// U32RangeCheckGate (u32/src/gates/range_check_u32.rs:89-111)
Program u32_range_check_gate(unsigned num_input_limbs, ImmediatePool &pool) {
    GateAsm g(&pool);
    for (unsigned i = 0; i < num_input_limbs; i++) {
        g.release();
        std::vector<int> small, aux;
        for (int k = 0; k < 4; k++) small.push_back(g.imm(k));
        for (unsigned j = 0; j < 16; j++) aux.push_back(g.wire(num_input_limbs + 16 * i + j));
        const int accu = g.reduce_with_powers(aux, 4);
        const int inp = g.wire(i);
        g.emit(g.sub(accu, inp, accu));
        g.free({accu, inp});
        for (int a : aux) {
            const int p = g.range_product(a, small);
            g.emit(p);
            g.free1(p);
        }
    }
    return g.instrs;
}

// ComparisonGate (u32/src/gates/comparison.rs:325-402)
Program comparison_gate(unsigned num_bits, unsigned num_chunks, ImmediatePool &pool) {
    GateAsm g(&pool);
    const unsigned cb = (num_bits + num_chunks - 1) / num_chunks, nc = num_chunks;
    for (unsigned which = 0; which < 2; which++) {
        std::vector<int> ws;
        for (unsigned i = 0; i < nc; i++) ws.push_back(4 + which * nc + i);
        const int accu = g.reduce_with_powers(ws, 1ull << cb, true);
        const int inp = g.wire(which);
        g.emit(g.sub(accu, inp, accu));
        g.free({accu, inp});
    }
    g.release();
    if (cb > 4) throw std::runtime_error("comparison chunks wider than 4 bits are not supported by this emitter");
    std::vector<int> small;
    for (unsigned k = 0; k < (1u << cb); k++) small.push_back(g.imm(k));
    const int base = g.imm(1ull << cb);
    const int one = small[1];
    int msd = g.imm(0);
    for (unsigned i = 0; i < nc; i++) {
        const int f = g.wire(4 + i), s2 = g.wire(4 + nc + i);
        int p = g.range_product(f, small);
        g.emit(p);
        g.free1(p);
        p = g.range_product(s2, small);
        g.emit(p);
        g.free1(p);
        const int diff = g.sub(s2, f);
        const int dummy = g.wire(4 + 2 * nc + i), eq = g.wire(4 + 3 * nc + i), inter = g.wire(4 + 4 * nc + i);
        const int neq = g.sub(one, eq);
        const int t = g.mul(diff, dummy);
        g.emit(g.sub(t, neq, t));
        g.emit(g.mul(eq, diff, t));
        g.mul(eq, msd, t);
        g.emit(g.sub(inter, t, t));
        g.mul(neq, diff, t);
        g.add(inter, t, msd);
        g.free({f, s2, diff, dummy, eq, inter, neq, t});
    }
    const int w3 = g.wire(3);
    g.emit(g.sub(w3, msd, msd));
    std::vector<int> bits;
    for (unsigned i = 0; i < cb + 1; i++) bits.push_back(g.wire(4 + 5 * nc + i));
    for (int b : bits) {
        const int t = g.sub(one, b);
        g.emit(g.mul(b, t, t));
        g.free1(t);
    }
    const int comb = g.reduce_with_powers(bits, 2);
    const int t = g.add(w3, base);
    g.emit(g.sub(t, comb, t));
    const int rb = g.wire(2);
    g.emit(g.sub(rb, bits[cb], t));
    return g.instrs;
}

// RandomAccessGate (plonky2/src/gates/random_access.rs:409-450)
Program random_access_gate(unsigned bits, unsigned num_copies, unsigned num_extra_constants, ImmediatePool &pool) {
    GateAsm g(&pool);
    const unsigned vs = 1u << bits;
    const unsigned routed = (2 + vs) * num_copies + num_extra_constants;
    for (unsigned c = 0; c < num_copies; c++) {
        g.release();
        const unsigned o = (2 + vs) * c;
        const int one = g.imm(1);
        std::vector<int> bs;
        for (unsigned i = 0; i < bits; i++) bs.push_back(g.wire(routed + c * bits + i));
        for (int b : bs) {
            const int t = g.sub(b, one);
            g.emit(g.mul(b, t, t));
            g.free1(t);
        }
        const int rec = g.imm(0);
        for (size_t k = bs.size(); k-- > 0;) {
            g.add(rec, rec, rec);
            g.add(rec, bs[k], rec);
        }
        const int idx = g.wire(o);
        g.emit(g.sub(rec, idx, rec));
        g.free({rec, idx});
        std::vector<int> items;
        for (unsigned i = 0; i < vs; i++) items.push_back(g.wire(o + 2 + i));
        for (int b : bs) {
            std::vector<int> nxt;
            for (size_t k = 0; k < items.size() / 2; k++) {
                const int x = items[2 * k], y = items[2 * k + 1];
                g.sub(y, x, y);
                g.mul(b, y, y);
                g.add(x, y, x);
                g.free1(y);
                nxt.push_back(x);
            }
            items = nxt;
        }
        const int claimed = g.wire(o + 1);
        g.emit(g.sub(items[0], claimed, claimed));
    }
    g.release();
    for (unsigned i = 0; i < num_extra_constants; i++) {
        const int c = g.constant(i), w = g.wire((2 + vs) * num_copies + i);
        g.emit(g.sub(c, w, c));
        g.free({c, w});
    }
    return g.instrs;
}

// PoseidonGate (plonky2/src/gates/poseidon.rs:485-564) over the permutation's own tables (hash/poseidon.rs:53-151,
// poseidon_goldilocks.rs:21-212)
Program poseidon_gate(ImmediatePool &pool) {
    GateAsm g(&pool);
    constexpr int SW = 12, WIRE_SWAP = 24, START_DELTA = 25;
    constexpr int START_FULL_0 = START_DELTA + 4, START_PARTIAL = START_FULL_0 + SW * 3, START_FULL_1 = START_PARTIAL + 22;
    const int one = g.imm(1);
    const int swap = g.wire(WIRE_SWAP);
    {
        const int t = g.sub(swap, one);
        g.emit(g.mul(swap, t, t));
        g.free({t, one});
    }
    int state[SW];
    for (int i = 0; i < 4; i++) {
        const int lhs = g.wire(i), rhs = g.wire(i + 4), delta = g.wire(START_DELTA + i);
        const int t = g.sub(rhs, lhs);
        g.mul(swap, t, t);
        g.emit(g.sub(t, delta, t));
        g.free1(t);
        state[i] = g.add(lhs, delta, lhs);
        state[i + 4] = g.sub(rhs, delta, rhs);
        g.free1(delta);
    }
    g.free1(swap);
    for (int i = 8; i < SW; i++) state[i] = g.wire(i);

    auto constant_layer = [&](int rc) {
        for (int i = 0; i < SW; i++) {
            const int c = g.imm(POSEIDON_ALL_ROUND_CONSTANTS[rc * SW + i]);
            g.add(state[i], c, state[i]);
            g.free1(c);
        }
    };
    auto sbox = [&](int x) {
        const int x2 = g.mul(x, x);
        const int x4 = g.mul(x2, x2);
        g.mul(x, x2, x2);
        g.mul(x2, x4, x);
        g.free({x2, x4});
    };
    auto mds_layer = [&]() {
        // row r = MDS_DIAG[r] * state[r] + sum_i MDS_CIRC[i] * state[(i + r) % 12]: thirteen weights below 64
        int fresh[SW];
        for (int r = 0; r < SW; r++) {
            std::vector<std::pair<int, uint64_t>> terms;
            for (int i = 0; i < SW; i++) terms.push_back({state[(i + r) % SW], POSEIDON_MDS_CIRC[i]});
            if (POSEIDON_MDS_DIAG[r]) terms.push_back({state[r], POSEIDON_MDS_DIAG[r]});
            fresh[r] = g.weighted_sum(terms);
        }
        for (int i = 0; i < SW; i++) g.free1(state[i]);
        for (int i = 0; i < SW; i++) state[i] = fresh[i];
    };
    auto check_against_wire = [&](int i, int wire) {
        const int sin = g.wire(wire);
        g.emit(g.sub(state[i], sin, state[i]));
        g.free1(state[i]);
        state[i] = sin;
    };

    int rc = 0;
    for (int r = 0; r < 4; r++) {
        constant_layer(rc);
        if (r != 0)
            for (int i = 0; i < SW; i++) check_against_wire(i, START_FULL_0 + SW * (r - 1) + i);
        for (int i = 0; i < SW; i++) sbox(state[i]);
        mds_layer();
        rc++;
    }
    // partial_first_constant_layer + mds_partial_layer_init
    for (int i = 0; i < SW; i++) {
        const int c = g.imm(POSEIDON_FAST_PARTIAL_FIRST_ROUND_CONSTANT[i]);
        g.add(state[i], c, state[i]);
        g.free1(c);
    }
    {
        int fresh[SW];
        fresh[0] = state[0];
        for (int c = 1; c < SW; c++) {
            const int accu = g.imm(0);
            for (int r = 1; r < SW; r++) {
                const int m = g.imm(POSEIDON_FAST_PARTIAL_ROUND_INITIAL_MATRIX[(r - 1) * 11 + (c - 1)]);
                g.mul(state[r], m, m);
                g.add(accu, m, accu);
                g.free1(m);
            }
            fresh[c] = accu;
        }
        for (int i = 1; i < SW; i++) g.free1(state[i]);
        for (int i = 0; i < SW; i++) state[i] = fresh[i];
    }
    for (int r = 0; r < 22; r++) {
        check_against_wire(0, START_PARTIAL + r);
        sbox(state[0]);
        if (r < 21) {
            const int c = g.imm(POSEIDON_FAST_PARTIAL_ROUND_CONSTANTS[r]);
            g.add(state[0], c, state[0]);
            g.free1(c);
        }
        // mds_partial_layer_fast
        const int k = g.imm((u128)POSEIDON_MDS_CIRC[0] + POSEIDON_MDS_DIAG[0]);
        const int d = g.mul(state[0], k);
        g.free1(k);
        for (int i = 1; i < SW; i++) {
            const int wh = g.imm(POSEIDON_FAST_PARTIAL_ROUND_W_HATS[r * 11 + i - 1]);
            g.mul(state[i], wh, wh);
            g.add(d, wh, d);
            g.free1(wh);
        }
        for (int i = 1; i < SW; i++) {
            const int v = g.imm(POSEIDON_FAST_PARTIAL_ROUND_VS[r * 11 + i - 1]);
            g.mul(state[0], v, v);
            g.add(state[i], v, state[i]);
            g.free1(v);
        }
        g.free1(state[0]);
        state[0] = d;
    }
    rc += 22;
    for (int r = 0; r < 4; r++) {
        constant_layer(rc);
        for (int i = 0; i < SW; i++) check_against_wire(i, START_FULL_1 + SW * r + i);
        for (int i = 0; i < SW; i++) sbox(state[i]);
        mds_layer();
        rc++;
    }
    for (int i = 0; i < SW; i++) {
        const int out = g.wire(SW + i);
        g.emit(g.sub(state[i], out, out));
        g.free1(out);
    }
    return g.instrs;
}

// ---- the gates of upstream plonky2 beyond the ed25519 list (round 5) -------------------------------------------------------------------
// Extension-field gates see pairs of wires as elements of F_p[X]/(X^2 - 7) (EvaluationVarsBase::get_local_ext, plonk/vars.rs:122-129;
// field/src/extension/quadratic.rs:173-185). A pair = (register of c0, register of c1). Every helper allocates its registers in the
// same order as its twin in gate_program.py: the two emitters must agree instruction for instruction.
struct Ext {
    int c0, c1;
    int operator[](int k) const { return k ? c1 : c0; }
};
Ext ext_wire(GateAsm &g, unsigned at) {
    const int a = g.wire((int)at);
    const int b = g.wire((int)at + 1);
    return Ext{a, b};
}
// (x0 + x1 X)(y0 + y1 X): c0 = x0 y0 + 7 x1 y1 through an accumulator (weights 1 and 7, one fold), c1 = x0 y1 + x1 y0
Ext ext_mul(GateAsm &g, Ext x, Ext y) {
    const int t00 = g.mul(x.c0, y.c0);
    const int t11 = g.mul(x.c1, y.c1);
    const int c0 = g.weighted_sum({{t00, 1}, {t11, 7}});
    g.free({t00, t11});
    const int t01 = g.mul(x.c0, y.c1);
    const int t10 = g.mul(x.c1, y.c0);
    const int c1 = g.add(t01, t10, t01);
    g.free1(t10);
    return Ext{c0, c1};
}
void ext_free(GateAsm &g, Ext x) { g.free({x.c0, x.c1}); }
// emit a - b component by component (to_basefield_array); `a` is overwritten
void ext_emit_diff(GateAsm &g, Ext a, Ext b) {
    g.emit(g.sub(a.c0, b.c0, a.c0));
    g.emit(g.sub(a.c1, b.c1, a.c1));
}

// ArithmeticExtensionGate { num_ops } (plonky2/src/gates/arithmetic_extension.rs:129-147)
Program arithmetic_extension_gate(unsigned num_ops, ImmediatePool &pool) {
    GateAsm g(&pool);
    for (unsigned i = 0; i < num_ops; i++) {
        g.release();
        const int c0 = g.constant(0), c1 = g.constant(1);
        const Ext m0 = ext_wire(g, 8 * i), m1 = ext_wire(g, 8 * i + 2), ad = ext_wire(g, 8 * i + 4), out = ext_wire(g, 8 * i + 6);
        const Ext m = ext_mul(g, m0, m1);
        for (int k = 0; k < 2; k++) {  // computed = m * c0 + addend * c1
            g.mul(m[k], c0, m[k]);
            g.mul(ad[k], c1, ad[k]);
            g.add(m[k], ad[k], m[k]);
        }
        ext_emit_diff(g, out, m);
    }
    return g.instrs;
}

// MulExtensionGate { num_ops } (plonky2/src/gates/multiplication_extension.rs:122-137)
Program mul_extension_gate(unsigned num_ops, ImmediatePool &pool) {
    GateAsm g(&pool);
    for (unsigned i = 0; i < num_ops; i++) {
        g.release();
        const int c0 = g.constant(0);
        const Ext m0 = ext_wire(g, 6 * i), m1 = ext_wire(g, 6 * i + 2), out = ext_wire(g, 6 * i + 4);
        const Ext m = ext_mul(g, m0, m1);
        for (int k = 0; k < 2; k++) g.mul(m[k], c0, m[k]);
        ext_emit_diff(g, out, m);
    }
    return g.instrs;
}

// ReducingGate { num_coeffs } (plonky2/src/gates/reducing.rs:160-181) and, with extension_coeffs, ReducingExtensionGate
// (reducing_extension.rs:157-178): acc_i = acc_{i-1} * alpha + coeff_i, the last accumulator being the output wires
Program reducing_gate(unsigned num_coeffs, ImmediatePool &pool, bool extension_coeffs) {
    GateAsm g(&pool);
    const unsigned d = 2, start_coeffs = 3 * d, start_accs = start_coeffs + (extension_coeffs ? d * num_coeffs : num_coeffs);
    const Ext alpha = ext_wire(g, d);
    Ext acc = ext_wire(g, 2 * d);
    for (unsigned i = 0; i < num_coeffs; i++) {
        const Ext t = ext_mul(g, acc, alpha);
        ext_free(g, acc);
        if (extension_coeffs) {
            const Ext c = ext_wire(g, start_coeffs + d * i);
            g.add(t.c0, c.c0, t.c0);
            g.add(t.c1, c.c1, t.c1);
            ext_free(g, c);
        } else {
            const int c = g.wire((int)(start_coeffs + i));
            g.add(t.c0, c, t.c0);
            g.free1(c);
        }
        const Ext nxt = ext_wire(g, i == num_coeffs - 1 ? 0 : start_accs + d * i);
        ext_emit_diff(g, t, nxt);
        ext_free(g, t);
        acc = nxt;
    }
    return g.instrs;
}

// ExponentiationGate { num_power_bits } (plonky2/src/gates/exponentiation.rs:266-298)
Program exponentiation_gate(unsigned n, ImmediatePool &pool) {
    GateAsm g(&pool);
    const int one = g.imm(1);
    const int base = g.wire(0);
    int prev = -1;
    for (unsigned i = 0; i < n; i++) {
        const int cur_bit = g.wire((int)(1 + (n - 1 - i)));  // power bits are little-endian, accumulated big-endian
        const int t = g.mul(cur_bit, base);
        g.add(t, one, t);
        g.sub(t, cur_bit, t);  // cur_bit * base + (1 - cur_bit)
        g.free1(cur_bit);
        if (prev >= 0) {
            g.mul(prev, prev, prev);
            g.mul(prev, t, t);
            g.free1(prev);
        }
        const int inter = g.wire((int)(2 + n + i));
        g.emit(g.sub(t, inter, t));
        g.free1(t);
        prev = inter;
    }
    const int out = g.wire((int)(1 + n));
    g.emit(g.sub(out, prev, out));
    return g.instrs;
}

// PoseidonMdsGate (plonky2/src/gates/poseidon_mds.rs:184-204): outputs = mds_layer_field(inputs) over F_p^2 — every component a sum of
// thirteen small multiples, one accumulator fold each
Program poseidon_mds_gate(ImmediatePool &pool) {
    GateAsm g(&pool);
    constexpr int SW = 12;
    Ext ins[SW];
    for (int i = 0; i < SW; i++) ins[i] = ext_wire(g, 2 * i);
    for (int r = 0; r < SW; r++)
        for (int k = 0; k < 2; k++) {
            std::vector<std::pair<int, uint64_t>> terms;
            for (int i = 0; i < SW; i++) terms.push_back({ins[(i + r) % SW][k], POSEIDON_MDS_CIRC[i]});
            if (POSEIDON_MDS_DIAG[r]) terms.push_back({ins[r][k], POSEIDON_MDS_DIAG[r]});
            const int computed = g.weighted_sum(terms);
            const int out = g.wire(2 * (SW + r) + k);
            g.emit(g.sub(out, computed, out));
            g.free({out, computed});
        }
    return g.instrs;
}

uint64_t mulmod(uint64_t a, uint64_t b) { return (uint64_t)((u128)a * b % P); }
uint64_t root_of_unity(unsigned bits) {  // F::primitive_root_of_unity (field/src/types.rs:268-272)
    uint64_t r = 1753635133440165772ull;
    for (unsigned i = bits; i < 32; i++) r = mulmod(r, r);
    return r;
}

// HighDegreeInterpolationGate (plonky2/src/gates/high_degree_interpolation.rs:119-147) / LowDegreeInterpolationGate
// (low_degree_interpolation.rs:356-404); wire layout gates/interpolation.rs:19-76
Program interpolation_gate(unsigned subgroup_bits, ImmediatePool &pool, bool low_degree) {
    GateAsm g(&pool);
    const unsigned d = 2, np = 1u << subgroup_bits;
    const unsigned start_values = 1, eval_point = 1 + np * d, eval_value = eval_point + d, start_coeffs = eval_value + d, end_coeffs = start_coeffs + np * d;
    const uint64_t w = root_of_unity(subgroup_bits);
    const int shift = g.wire(0);
    // Horner with a base-field point in register x: acc = acc * x + c, component-wise; returns a fresh pair
    auto eval_base = [&](const std::vector<Ext> &cs, int x) {
        const int z = g.imm(0);
        const int a0 = g.add(cs.back().c0, z);
        const int a1 = g.add(cs.back().c1, z);  // copies of the leading coefficient (ADD with a zero immediate)
        g.free1(z);
        const Ext acc{a0, a1};
        for (size_t i = cs.size() - 1; i-- > 0;)
            for (int k = 0; k < 2; k++) {
                g.mul(acc[k], x, acc[k]);
                g.add(acc[k], cs[i][k], acc[k]);
            }
        return acc;
    };
    std::vector<Ext> coeffs;
    for (unsigned i = 0; i < np; i++) coeffs.push_back(ext_wire(g, start_coeffs + d * i));
    if (low_degree) {
        // powers of the shift: wire i (i = 2..np-1) must be shift^(i-1) * shift; coefficient i is altered by shift^i on the way
        int prev = shift;
        for (unsigned i = 1; i < np; i++) {
            for (int k = 0; k < 2; k++) g.mul(coeffs[i][k], prev, coeffs[i][k]);
            if (i < np - 1) {
                const int nxt = g.wire((int)(end_coeffs + i - 1));
                const int t = g.mul(prev, shift);
                g.emit(g.sub(t, nxt, t));
                g.free1(t);
                if (prev != shift) g.free1(prev);
                prev = nxt;
            }
        }
        if (prev != shift) g.free1(prev);
    }
    uint64_t wi = 1;
    for (unsigned i = 0; i < np; i++) {
        const int x = g.imm(wi);
        wi = mulmod(wi, w);
        if (!low_degree) g.mul(x, shift, x);  // coset(shift) = g^i * shift
        const Ext computed = eval_base(coeffs, x);
        g.free1(x);
        const Ext value = ext_wire(g, start_values + d * i);
        ext_emit_diff(g, value, computed);
        ext_free(g, value);
        ext_free(g, computed);
    }
    const Ext ep = ext_wire(g, eval_point);
    Ext acc;
    if (low_degree) {
        for (const Ext &c : coeffs) ext_free(g, c);
        Ext prev = ep;
        bool prev_is_ep = true;
        for (unsigned i = 1; i < np - 1; i++) {  // powers of the evaluation point: wire pair i+1 must be (pair i) * point
            const Ext nxt = ext_wire(g, end_coeffs + np - 2 + (i - 1) * d);
            const Ext t = ext_mul(g, prev, ep);
            ext_emit_diff(g, t, nxt);
            ext_free(g, t);
            if (!prev_is_ep) ext_free(g, prev);
            prev = nxt;
            prev_is_ep = false;
        }
        if (!prev_is_ep) ext_free(g, prev);
        acc = ext_wire(g, start_coeffs);  // eval_with_powers uses the ORIGINAL coefficients: c_0 + sum c_i * point^i
        for (unsigned i = 1; i < np; i++) {
            const Ext pw = i == 1 ? ep : ext_wire(g, end_coeffs + np - 2 + (i - 2) * d);
            const Ext c = ext_wire(g, start_coeffs + d * i);
            const Ext t = ext_mul(g, pw, c);
            g.add(acc.c0, t.c0, acc.c0);
            g.add(acc.c1, t.c1, acc.c1);
            ext_free(g, t);
            ext_free(g, c);
            if (i != 1) ext_free(g, pw);
        }
    } else {
        acc = coeffs.back();  // interpolant.eval(point): Horner over the extension
        for (size_t i = coeffs.size() - 1; i-- > 0;) {
            const Ext t = ext_mul(g, acc, ep);
            g.add(t.c0, coeffs[i].c0, t.c0);
            g.add(t.c1, coeffs[i].c1, t.c1);
            ext_free(g, acc);
            acc = t;
        }
    }
    const Ext value = ext_wire(g, eval_value);
    ext_emit_diff(g, value, acc);
    return g.instrs;
}

// Sane parameter ranges per kind, checked BEFORE anything is emitted or allocated: every wire / constant index of the program must
// fit the 16-bit instruction fields (GateAsm::field throws as the backstop) and nothing may grow without limit. The largest real
// gates are far inside (ed25519: arithmetic 20 ops, base_sum 63 limbs, u32_add_many 16 addends x 4 ops, random_access 4 bits).
void check_params(const GlGateSpec &s) {
    const uint32_t *p = s.params;
    auto need = [&](bool ok, const char *what) {
        if (!ok) throw std::invalid_argument(std::string("gate kind ") + std::to_string(s.kind) + ": " + what);
    };
    switch (s.kind) {
        case GL_GATE_CONSTANT: need(p[0] >= 1 && p[0] <= 4096, "num_consts must be in 1..4096"); break;
        case GL_GATE_ARITHMETIC: need(p[0] >= 1 && p[0] <= 4096, "num_ops must be in 1..4096"); break;
        case GL_GATE_BASE_SUM: need(p[0] >= 2 && p[0] <= 256, "base B must be in 2..256"); need(p[1] >= 1 && p[1] <= 64, "num_limbs must be in 1..64"); break;
        case GL_GATE_U32_ADD_MANY: need(p[0] <= 256, "num_addends must be at most 256"); need(p[1] >= 1 && p[1] <= 256, "num_ops must be in 1..256"); break;
        case GL_GATE_U32_ARITHMETIC:
        case GL_GATE_U32_SUBTRACTION: need(p[0] >= 1 && p[0] <= 1024, "num_ops must be in 1..1024"); break;
        case GL_GATE_U32_RANGE_CHECK: need(p[0] <= 1024, "num_input_limbs must be at most 1024"); break;
        case GL_GATE_COMPARISON: need(p[0] >= 1 && p[0] <= 64, "num_bits must be in 1..64"); need(p[1] >= 1 && p[1] <= p[0], "num_chunks must be in 1..num_bits"); break;
        case GL_GATE_RANDOM_ACCESS:
            need(p[0] <= 10, "bits must be at most 10");
            need(p[1] >= 1 && p[1] <= 256, "num_copies must be in 1..256");
            need(p[2] <= 256, "num_extra_constants must be at most 256");
            break;
        case GL_GATE_ARITHMETIC_EXTENSION:
        case GL_GATE_MUL_EXTENSION: need(p[0] >= 1 && p[0] <= 2048, "num_ops must be in 1..2048"); break;
        case GL_GATE_REDUCING:
        case GL_GATE_REDUCING_EXTENSION: need(p[0] >= 1 && p[0] <= 4096, "num_coeffs must be in 1..4096"); break;
        case GL_GATE_EXPONENTIATION: need(p[0] >= 1 && p[0] <= 4096, "num_power_bits must be in 1..4096"); break;
        case GL_GATE_LOW_DEGREE_INTERPOLATION:
        case GL_GATE_HIGH_DEGREE_INTERPOLATION: need(p[0] >= 1 && p[0] <= 4, "subgroup_bits must be in 1..4"); break;
        default: break;
    }
}

Program build_gate(const GlGateSpec &s, ImmediatePool &pool) {
    const uint32_t *p = s.params;
    check_params(s);
    switch (s.kind) {
        case GL_GATE_NOOP: return Program();
        case GL_GATE_CONSTANT: return constant_gate(p[0]);
        case GL_GATE_PUBLIC_INPUT: return public_input_gate();
        case GL_GATE_ARITHMETIC: return arithmetic_gate(p[0]);
        case GL_GATE_BASE_SUM: return base_sum_gate(p[0], p[1], pool);
        case GL_GATE_U32_ADD_MANY: return u32_add_many_gate(p[0], p[1], pool);
        case GL_GATE_U32_ARITHMETIC: return u32_arithmetic_gate(p[0], pool);
        case GL_GATE_U32_SUBTRACTION: return u32_subtraction_gate(p[0], pool);
        case GL_GATE_U32_RANGE_CHECK: return u32_range_check_gate(p[0], pool);
        case GL_GATE_COMPARISON: return comparison_gate(p[0], p[1], pool);
        case GL_GATE_RANDOM_ACCESS: return random_access_gate(p[0], p[1], p[2], pool);
        case GL_GATE_POSEIDON: return poseidon_gate(pool);
        case GL_GATE_ARITHMETIC_EXTENSION: return arithmetic_extension_gate(p[0], pool);
        case GL_GATE_MUL_EXTENSION: return mul_extension_gate(p[0], pool);
        case GL_GATE_REDUCING: return reducing_gate(p[0], pool, false);
        case GL_GATE_REDUCING_EXTENSION: return reducing_gate(p[0], pool, true);
        case GL_GATE_EXPONENTIATION: return exponentiation_gate(p[0], pool);
        case GL_GATE_POSEIDON_MDS: return poseidon_mds_gate(pool);
        case GL_GATE_LOW_DEGREE_INTERPOLATION: return interpolation_gate(p[0], pool, true);
        case GL_GATE_HIGH_DEGREE_INTERPOLATION: return interpolation_gate(p[0], pool, false);
        default: throw std::runtime_error("no register-program emitter for gate kind " + std::to_string(s.kind));
    }
}

GlError emit_error(const std::string &msg) {
    GlError e;
    e.code = -1;
    e.message = strdup(msg.c_str());
    return e;
}

}  // namespace

extern "C" GlError gl_gate_programs_emit(const GlGateSpec *gates, uint32_t num_gates, const uint32_t *group_bounds, uint32_t num_selectors,
                                         GlGatePrograms *out) {
    if (!out || (num_gates && !gates) || (num_selectors && !group_bounds)) return emit_error("gl_gate_programs_emit: null pointer");
    memset(out, 0, sizeof *out);
    try {
        ImmediatePool pool;
        std::vector<GlGateInstr> instrs;
        std::vector<GlGateDesc> descs;
        uint32_t max_constraints = 0;
        for (uint32_t row = 0; row < num_gates; row++) {
            const uint32_t si = gates[row].selector_index;
            if (si >= num_selectors) return emit_error("gl_gate_programs_emit: selector_index out of range");
            const Program prog = build_gate(gates[row], pool);
            uint32_t emits = 0;
            for (const GlGateInstr &in : prog) emits += in.op == EMIT;
            max_constraints = std::max(max_constraints, emits);
            descs.push_back(GlGateDesc{row, si, group_bounds[2 * si], group_bounds[2 * si + 1], (uint32_t)instrs.size(), (uint32_t)prog.size()});
            instrs.insert(instrs.end(), prog.begin(), prog.end());
        }
        if (instrs.empty()) instrs.push_back(GlGateInstr{0, 0, 0, 0});  // as gate_program.pack_program: never an empty array
        out->num_instrs = (uint32_t)instrs.size();
        out->num_gates = (uint32_t)descs.size();
        out->num_immediates = (uint32_t)pool.values.size();
        out->num_gate_constraints = max_constraints;
        out->instrs = (GlGateInstr *)malloc(std::max<size_t>(1, instrs.size()) * sizeof(GlGateInstr));
        out->gates = (GlGateDesc *)malloc(std::max<size_t>(1, descs.size()) * sizeof(GlGateDesc));
        out->immediates = (uint64_t *)malloc(std::max<size_t>(1, pool.values.size()) * sizeof(uint64_t));
        if (!out->instrs || !out->gates || !out->immediates) {
            gl_gate_programs_free(out);
            return emit_error("gl_gate_programs_emit: out of memory");
        }
        memcpy(out->instrs, instrs.data(), instrs.size() * sizeof(GlGateInstr));
        if (!descs.empty()) memcpy(out->gates, descs.data(), descs.size() * sizeof(GlGateDesc));
        if (!pool.values.empty()) memcpy(out->immediates, pool.values.data(), pool.values.size() * sizeof(uint64_t));
    } catch (const std::invalid_argument &ex) {
        gl_gate_programs_free(out);
        GlError e = emit_error(std::string("gl_gate_programs_emit: ") + ex.what());
        e.code = GL_E_INVALID;
        return e;
    } catch (const std::exception &ex) {
        gl_gate_programs_free(out);
        return emit_error(std::string("gl_gate_programs_emit: ") + ex.what());
    }
    GlError ok;
    ok.code = 0;
    ok.message = nullptr;
    return ok;
}

extern "C" void gl_gate_programs_free(GlGatePrograms *p) {
    if (!p) return;
    free(p->instrs);
    free(p->gates);
    free(p->immediates);
    memset(p, 0, sizeof *p);
}
