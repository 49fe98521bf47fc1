// gate_jit.hip — the circuit-specific part of compute_quotient_polys as a RUN-TIME COMPILED kernel.
//
// The reference hard-wires the 25 gates of one circuit into its quotient kernel
// (cuda/plonky2_gpu_impl.cuh:600-685). Here a circuit's gates arrive as register programs
// (GlGateInstr, include/plonky2_hip.h) and are turned into straight-line HIP source — one
// __noinline__ device function per gate, registers as local variables, immediates as literals —
// compiled for gfx950 with hiprtc when the circuit is built, and launched as an ordinary kernel:
// the program's registers live in VGPRs and the instruction stream is real machine code, instead of
// an interpreter that keeps 64 registers in scratch memory and decodes an opcode per operation.
//
// What the kernel computes for the LDE point held by leaf t
// (evaluate_gate_constraints_base_batch, plonky2/src/plonk/vanishing_poly.rs:267-306; Gate::eval_filtered,
// gates/gate.rs:86-109; compute_filter, gates/gate.rs:261-268):
//     G_c(t) = sum_k alpha_c^k * sum_g filter_g(t) * constraint_{g,k}(t)
// i.e. the gate-constraint tail of reduce_with_powers_multi (plonk_common.rs:97-114), which
// quotient_values_kernel then continues through the permutation terms. Powers of alpha come from a
// small device table read with wave-uniform (scalar) loads.
#include "gate_jit.h"

#include <hip/hiprtc.h>

#include <dlfcn.h>
#include <signal.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <time.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <algorithm>
#include <functional>
#include <map>
#include <mutex>
#include <sstream>
#include <thread>
#include <vector>

#include "gl_field.h"
#include "knobs.h"

namespace plonky2_hip {

namespace {
const char *const GL_FIELD_SRC =
#include "build/gl_field_src.inc"
    ;
const char *const GL_JIT_FIELD_SRC =
#include "build/gate_jit_field_src.inc"
    ;

enum : uint16_t { GP_LOAD_WIRE, GP_LOAD_CONST, GP_LOAD_PI, GP_LOAD_IMM, GP_ADD, GP_SUB, GP_MUL, GP_EMIT, GP_MULK, GP_ACC, GP_ACCR };
constexpr uint32_t MAX_REGS = 64, MAX_CH = 4;
const char *const JIT_ARCH = "gfx950";

// Where compiled code objects are kept: $PLONKY2_HIP_KERNEL_CACHE (empty = no cache), else the directory
// `kernel_cache` next to this shared library if it exists — the place __graft_entry__.build() precompiles the
// compiled-in ed25519 gate kernel into, so that it ships with the library like any ahead-of-time kernel.
std::string kernel_cache_dir() {
    if (const char *dir = getenv("PLONKY2_HIP_KERNEL_CACHE")) return dir;
    Dl_info info;
    if (!dladdr(reinterpret_cast<const void *>(&kernel_cache_dir), &info) || !info.dli_fname) return "";
    std::string lib(info.dli_fname);
    size_t slash = lib.rfind('/');
    std::string dir = (slash == std::string::npos ? std::string(".") : lib.substr(0, slash)) + "/kernel_cache";
    struct stat st;
    return (stat(dir.c_str(), &st) == 0 && S_ISDIR(st.st_mode)) ? dir : "";
}
}  // namespace

// A circuit's gates are compiled as several UNITS — each a hiprtc program of its own with some of the gates and a kernel
// that adds their share to the output — so that the units compile side by side (the ed25519 table: 56 s of compilation, 8.5 of
// them the Poseidon gate; eight units on eight cores take about as long as the largest) and a unit that another circuit
// shares comes out of the on-disk cache. The units' kernels run one after the other on the caller's stream; every one is bound
// by the vector ALU's issue rate, so their sum takes what the single kernel took.
struct GateUnit {
    std::vector<uint32_t> gates;  // indices into the circuit's gate list
    std::string source;
    std::vector<char> code;
    hipModule_t module = nullptr;
    hipFunction_t fn = nullptr;
    uint64_t *d_tab = nullptr;  // the module's g_tab: par[6] (where the LDE lives) | pih[4] | apow[num_challenges][num_constraints] (x 2 when fused) | bias
    // Constraints the generated code emits with a known constant added (the peephole pass below): bias[i] lists (k, b) for the
    // i-th gate of the unit — the code accumulates alpha^k (c_k + b), the launch supplies sum alpha^k b in g_bias to take off.
    std::vector<std::vector<std::pair<uint32_t, uint64_t>>> bias;
    bool biased = false;  // any gate of the unit has a bias
    std::string error;
};

struct GateKernel {
    std::vector<GateUnit> units;
    uint32_t num_challenges = 0, num_constraints = 0;
    bool fused = false;  // fused units read g_apow as pairs {alpha^k, alpha^k * 2^32}
    uint32_t wires_needed = 0, constants_needed = 0;  // 1 + the largest wire / constant column any gate loads
    std::string source;  // all units, for inspection
    // A launch rewrites the units' __constant__ tables (alpha powers, biases, public-inputs hash, LDE pointers) in stream order in
    // front of its kernels. Launches of ONE GateKernel from several streams / host threads (two proofs of one circuit in flight)
    // therefore take turns: the next launch's stream waits for the previous launch's kernels before its copies may overwrite what
    // they read. Launches of different GateKernels run concurrently. The host tables stay alive until the next launch has seen
    // the previous one's event, so that no launch has to drain its stream for the copies' sake.
    std::mutex launch_mu;
    hipEvent_t done = nullptr;
    bool launched = false;
    std::vector<std::vector<uint64_t>> h_tab;  // per unit: the host image of its GateTab
};

// What every consumer of gate programs checks before running them (the compiled kernel when it is generated, the
// interpreter's callers through gl_gate_program_validate): a gate must not emit more constraints than the circuit
// declares — the reference asserts "num_constraints() gave too low of a number" (plonk/vanishing_poly.rs:256-262,
// 296-303); silently dropping the surplus would leave those constraints unenforced — and the columns it loads must
// exist. Returns false and fills `error`.
bool gate_programs_validate(const uint16_t *instrs, uint32_t num_instrs, const uint32_t *gates, uint32_t num_gates, uint32_t num_imms,
                            uint32_t num_selectors, uint32_t ngc, uint32_t *wires_needed, uint32_t *constants_needed,
                            std::string *error) {
    uint32_t wn = 0, cn = num_selectors;
    if (num_gates == 0) {
        *error = "a gate table holds at least one gate";
        return false;
    }
    for (uint32_t g = 0; g < num_gates; g++) {
        const uint32_t *d = gates + 6 * g;
        const uint32_t si = d[1], gs = d[2], ge = d[3], ps = d[4], pl = d[5];
        if ((uint64_t)ps + pl > num_instrs || si >= num_selectors || gs > ge) {
            *error = "gate descriptor out of range";
            return false;
        }
        uint32_t emitted = 0;
        // One rule for the interpreter, the per-gate generator and the fused generator: a register is written before it is read, in
        // program order (a program that leans on registers starting at zero would build under one generator setting and not the other).
        bool written[MAX_REGS] = {};
        for (uint32_t pc = ps; pc < ps + pl; pc++) {
            const uint16_t op = instrs[4 * pc], dst = instrs[4 * pc + 1], a = instrs[4 * pc + 2], b = instrs[4 * pc + 3];
            const bool reads_a = op == GP_ADD || op == GP_SUB || op == GP_MUL || op == GP_EMIT || op == GP_MULK || op == GP_ACC;
            const bool reads_b = op == GP_ADD || op == GP_SUB || op == GP_MUL;
            if ((reads_a && !written[a & (MAX_REGS - 1)]) || (reads_b && !written[b & (MAX_REGS - 1)])) {
                *error = "gate " + std::to_string(g) + ": register read before any write (instruction " + std::to_string(pc - ps) + ")";
                return false;
            }
            if (op != GP_EMIT && op != GP_ACC && op <= GP_ACCR) written[dst & (MAX_REGS - 1)] = true;  // ACC's dst names an accumulator
            if (op == GP_LOAD_WIRE && (uint32_t)a + 1 > wn) wn = (uint32_t)a + 1;
            if (op == GP_LOAD_CONST && num_selectors + a + 1 > cn) cn = num_selectors + a + 1;
            if (op == GP_LOAD_IMM && a >= num_imms) {
                *error = "LOAD_IMM index out of range";
                return false;
            }
            if (op == GP_ACC && b >= num_imms) {
                *error = "ACC immediate index out of range";
                return false;
            }
            if (op == GP_EMIT) emitted++;
            if (op > GP_ACCR) {
                *error = "unknown opcode";
                return false;
            }
        }
        if (emitted > ngc) {
            *error = "gate " + std::to_string(g) + " emits " + std::to_string(emitted) + " constraints but num_gate_constraints is " +
                     std::to_string(ngc) + " (num_constraints() gave too low of a number)";
            return false;
        }
    }
    if (wires_needed) *wires_needed = wn;
    if (constants_needed) *constants_needed = cn;
    return true;
}

// ---- peephole pass ---------------------------------------------------------------------------------------------------------
// The programs are executed as written by the interpreter (plonk.hip) and by the oracles; the compiled kernel may compute the same
// field values more cheaply. Two rewrites, both found by following each register read back to the instruction that wrote the value
// (the programs are straight-line code):
//   1. ADD / SUB whose other operand is a LOAD_IMM below 2^32 become gl::add_small / gl::sub_small (two vector instructions instead of
//      four / five; adding or subtracting zero becomes a copy).
//   2. The range check of a base-4 limb, as every emitter writes it: t = a * b; u = t + 2; c = t * u; EMIT c, with t, u, c read nowhere
//      else. t (t + 2) = (t + 1)^2 - 1, and t + 1 is a * b + 1 at the price of a * b (gl::mul_add_small), so the ADD disappears: the
//      code emits (t + 1)^2 and the constant 1 it is too large by is taken off ONCE per gate and challenge, as sum of alpha^k over
//      those constraints (GateUnit::bias, g_bias) — wave-uniform, computed by the host with the powers of alpha.
//   3. Any other MUL / ADD / SUB with a LOAD_IMM operand reads the constant's halves as scalar operands (gl::mul_k, gl::add_k), and a
//      LOAD_IMM whose readers all do is not generated (two v_mov each).
// On the ed25519 table (1 838 such limbs among 21 467 operations) the executed vector instructions per LDE point go from 150 k to
// 135 k. PLONKY2_HIP_JIT_PEEPHOLE=0 turns the pass off (A/B, and the tests that hold one form against the other; like every switch of the
// generator it is read by the DIAGNOSTIC build of the library only, knobs.h).
struct Peep {
    enum Kind : uint8_t { PLAIN, ADD_SMALL, SUB_SMALL, COPY, MUL_ADD1, SQUARE, MUL_K, ADD_K, SKIP } kind = PLAIN;
    uint32_t src = 0;      // ADD_SMALL / SUB_SMALL / COPY / SQUARE / MUL_K / ADD_K: the register read
    int src_def = -1;      // ... and the instruction that wrote it
    uint64_t k = 0;        // ADD_SMALL / SUB_SMALL / MUL_K / ADD_K: the constant
    uint64_t bias = 0;     // EMIT: what the emitted value is too large by
};

static bool peephole_enabled() {
    const char *e = PLONKY2_KNOB("PLONKY2_HIP_JIT_PEEPHOLE");
    return !(e && e[0] == '0');
}

static std::vector<Peep> peephole(const uint16_t *instrs, uint32_t ps, uint32_t pl, const uint64_t *imms, uint32_t num_imms) {
    std::vector<Peep> out(pl);
    if (!peephole_enabled()) return out;
    // reaching definitions: which instruction wrote the value an operand reads (-1: none), and who reads each value
    std::vector<int> def_a(pl, -1), def_b(pl, -1);
    std::vector<std::vector<uint32_t>> uses(pl);
    int last[MAX_REGS];
    for (uint32_t r = 0; r < MAX_REGS; r++) last[r] = -1;
    auto op_of = [&](uint32_t i) { return instrs[4 * (ps + i)]; };
    auto fld = [&](uint32_t i, int f) { return instrs[4 * (ps + i) + f]; };
    for (uint32_t i = 0; i < pl; i++) {
        const uint16_t op = op_of(i);
        const bool reads_a = op == GP_ADD || op == GP_SUB || op == GP_MUL || op == GP_EMIT || op == GP_MULK || op == GP_ACC;
        const bool reads_b = op == GP_ADD || op == GP_SUB || op == GP_MUL;
        if (reads_a && (def_a[i] = last[fld(i, 2) & (MAX_REGS - 1)]) >= 0) uses[def_a[i]].push_back(i);
        if (reads_b && (def_b[i] = last[fld(i, 3) & (MAX_REGS - 1)]) >= 0) uses[def_b[i]].push_back(i);
        if (op != GP_EMIT && op != GP_ACC) last[fld(i, 1) & (MAX_REGS - 1)] = (int)i;  // ACC's dst names an accumulator
    }
    auto small_imm = [&](int d, uint64_t *v) {
        if (d < 0 || op_of(d) != GP_LOAD_IMM || fld(d, 2) >= num_imms) return false;
        *v = imms[fld(d, 2)] % glh::P;
        return *v <= 0xFFFFFFFFull;
    };
    for (uint32_t i = 0; i < pl; i++) {
        const uint16_t op = op_of(i);
        uint64_t v = 0;
        if (op == GP_ADD || op == GP_SUB) {
            const bool b_small = small_imm(def_b[i], &v);
            const bool a_small = !b_small && op == GP_ADD && small_imm(def_a[i], &v);
            if (!b_small && !a_small) continue;
            const int other = b_small ? def_a[i] : def_b[i];
            if (other < 0) continue;  // read before any write: left to the generator's own check
            out[i].kind = v == 0 ? Peep::COPY : op == GP_ADD ? Peep::ADD_SMALL : Peep::SUB_SMALL;
            out[i].src = fld(i, b_small ? 2 : 3) & (MAX_REGS - 1);
            out[i].src_def = other;
            out[i].k = v;
        }
    }
    for (uint32_t m2 = 0; m2 < pl; m2++) {
        if (op_of(m2) != GP_MUL || def_a[m2] < 0 || def_b[m2] < 0) continue;
        for (int swap = 0; swap < 2; swap++) {
            const int m1 = swap ? def_b[m2] : def_a[m2], ad = swap ? def_a[m2] : def_b[m2];
            if (m1 == ad || op_of(m1) != GP_MUL || op_of(ad) != GP_ADD) continue;
            // the ADD is t + 2 with t the very value of m1
            uint64_t v = 0;
            const bool t_is_a = def_a[ad] == m1 && small_imm(def_b[ad], &v) && v == 2;
            const bool t_is_b = !t_is_a && def_b[ad] == m1 && small_imm(def_a[ad], &v) && v == 2;
            if (!t_is_a && !t_is_b) continue;
            if (uses[m1].size() != 2 || uses[ad].size() != 1 || uses[m2].size() != 1) continue;  // (m1: the ADD and m2; each once)
            const uint32_t e = uses[m2][0];
            if (op_of(e) != GP_EMIT || def_a[m1] < 0 || def_b[m1] < 0) continue;
            if (out[m1].kind != Peep::PLAIN || out[m2].kind != Peep::PLAIN) continue;  // m1 already rewritten as another limb's square
            out[m1].kind = Peep::MUL_ADD1;
            out[ad].kind = Peep::SKIP;
            out[m2].kind = Peep::SQUARE;
            out[m2].src = fld(m1, 1) & (MAX_REGS - 1);
            out[m2].src_def = m1;
            out[e].bias = 1;
            break;
        }
    }
    // 3. what is left of MUL / ADD / SUB with a LOAD_IMM operand takes the constant as a scalar operand (gl::mul_k, gl::add_k; x - K is
    //    x + (p - K)), and a LOAD_IMM that nothing reads any more is not generated.
    auto any_imm = [&](int d, uint64_t *v) {
        if (d < 0 || op_of(d) != GP_LOAD_IMM || fld(d, 2) >= num_imms) return false;
        *v = imms[fld(d, 2)] % glh::P;
        return true;
    };
    for (uint32_t i = 0; i < pl; i++) {
        const uint16_t op = op_of(i);
        if ((op != GP_MUL && op != GP_ADD && op != GP_SUB) || out[i].kind != Peep::PLAIN || def_a[i] < 0 || def_b[i] < 0) continue;
        uint64_t v = 0;
        const bool b_imm = any_imm(def_b[i], &v);
        const bool a_imm = !b_imm && op != GP_SUB && any_imm(def_a[i], &v);
        if (!b_imm && !a_imm) continue;
        out[i].kind = op == GP_MUL ? Peep::MUL_K : Peep::ADD_K;
        out[i].src = fld(i, b_imm ? 2 : 3) & (MAX_REGS - 1);
        out[i].src_def = b_imm ? def_a[i] : def_b[i];
        out[i].k = op == GP_SUB ? (glh::P - v) % glh::P : v;
    }
    for (uint32_t d = 0; d < pl; d++) {
        if (op_of(d) != GP_LOAD_IMM) continue;
        bool read = false;
        for (uint32_t u : uses[d]) {
            const Peep::Kind kd = out[u].kind;
            const bool constant_only = kd == Peep::ADD_SMALL || kd == Peep::SUB_SMALL || kd == Peep::COPY || kd == Peep::MUL_K || kd == Peep::ADD_K;
            read |= !(kd == Peep::SKIP || (constant_only && out[u].src_def != (int)d));
        }
        if (!read) out[d].kind = Peep::SKIP;  // (also a LOAD_IMM nothing ever read)
    }
    return out;
}

// ---- fused units --------------------------------------------------------------------------------------------------------------
// One gate at a time, every gate loads the wires it reads and computes what it needs from them — and the gates of a circuit read the
// same wires and need the same things: the ten U32AddMany gates of the ed25519 table, U32Arithmetic, U32Subtraction and the range
// checks all range-check the limbs that sit in wires 60..233. Value numbering over the whole table leaves 1 861 of 5 107
// multiplications, 1 179 of 2 822 additions, 977 of 2 653 subtractions and 234 of 3 204 wire loads; and the loads are not free: a
// row's 1.9 KB of wires times the rows in flight does not stay in L2 between two gates, the gate kernels fetched 40 GB per quotient
// for a 3.9 GB LDE and ran 13.3 ms where the same arithmetic without loads runs 9.6 (profiles/r05_quotient_loads_experiment.txt).
//
// So the gates of a unit are generated as ONE straight-line function over a common value graph:
//   * every operation becomes a node keyed by what it computes (kind, operand nodes, constants; ADD and MUL with sorted operands;
//     an ACCR is the node "sum of weight_i * node_i"), so that two gates asking for the same value share the node;
//   * what stays per gate is where its constraints go: the alpha-accumulators (gl::DotCol2 per challenge) and the list
//     (constraint index k, node);
//   * the code is emitted in the order of the first gate's program, then the second's ..., skipping what exists — but every value is
//     PUSHED to its consumers the moment it is computed: an EMIT of any gate of the unit that takes it is issued right there
//     (its own k, its own accumulator), a term of an ACCR sum is accumulated right there, and an operation whose operands are now
//     all there is computed (and pushed in turn) IF that leads to an EMIT or to such a sum — otherwise it waits for the program
//     that wants it, and a wire it needs that was loaded long before is loaded again (fuse_schedule). A limb's range check is
//     computed once and lands in five gates' accumulators within a few instructions; nothing is kept for a later gate except the
//     running sums.
// What bounds a unit is the registers of those per-gate accumulators (12 per gate with two challenges, gl::DotCol2), hence few gates
// per unit (PLONKY2_HIP_JIT_FUSE_GATES, default 5), chosen by what they share (fuse_partition).
// PLONKY2_HIP_JIT_FUSE=0 generates one function per gate as before (A/B; tests hold the two against each other).
static bool fuse_enabled() {
    const char *e = PLONKY2_KNOB("PLONKY2_HIP_JIT_FUSE");
    return !(e && e[0] == '0');
}

// waves per SIMD the fused kernels are compiled for (their register budget: 128 VGPRs at 4, 168 at 3, 256 at 2)
static uint32_t fuse_waves() {
    if (const char *e = PLONKY2_KNOB("PLONKY2_HIP_JIT_WAVES")) {
        const long v = strtol(e, nullptr, 10);
        if (v >= 1 && v <= 8) return (uint32_t)v;
    }
    return 4;
}

// how many statements before its first use a wire is loaded
static uint32_t fuse_prefetch_distance() {
    if (const char *e = PLONKY2_KNOB("PLONKY2_HIP_JIT_PREFETCH")) {
        const long v = strtol(e, nullptr, 10);
        if (v >= 0 && v <= 4096) return (uint32_t)v;
    }
    return 16;
}

static uint32_t fuse_gates_per_unit() {
    if (const char *e = PLONKY2_KNOB("PLONKY2_HIP_JIT_FUSE_GATES")) {
        const long v = strtol(e, nullptr, 10);
        if (v >= 1 && v <= 64) return (uint32_t)v;
    }
    return 5;
}

struct FNode {
    enum Kind : uint8_t { WIRE, CONST, PI, IMM, ADD, SUB, MUL, MULK, ADD_SMALL, SUB_SMALL, MUL_ADD1, MUL_K, ADD_K, LIN } kind;
    uint32_t a = ~0u, b = ~0u;
    uint64_t k = 0;
    std::vector<std::pair<uint32_t, uint64_t>> terms;  // LIN: (node, weight < 2^32)
};

struct FGraph {
    std::vector<FNode> nodes;
    std::map<std::string, uint32_t> index;
    uint32_t intern(const FNode &n) {
        std::string key(1, (char)n.kind);
        auto put = [&](uint64_t v) { key.append(reinterpret_cast<const char *>(&v), sizeof v); };
        put(n.a), put(n.b), put(n.k);
        for (const auto &t : n.terms) put(t.first), put(t.second);
        auto it = index.find(key);
        if (it != index.end()) return it->second;
        nodes.push_back(n);
        index.emplace(std::move(key), (uint32_t)nodes.size() - 1);
        return (uint32_t)nodes.size() - 1;
    }
    uint32_t make(FNode::Kind kind, uint32_t a = ~0u, uint32_t b = ~0u, uint64_t k = 0) {
        FNode n;
        n.kind = kind, n.a = a, n.b = b, n.k = k;
        if ((kind == FNode::ADD || kind == FNode::MUL || kind == FNode::MUL_ADD1) && n.a > n.b) std::swap(n.a, n.b);
        return intern(n);
    }
};

struct FEmit {
    uint32_t k, node;
};

// One gate's program -> nodes of `g`, its EMITs in `emits`, the constants its emitted values are too large by in `bias`.
static bool fuse_gate(FGraph &g, const uint16_t *instrs, uint32_t ps, uint32_t pl, const uint64_t *imms, uint32_t num_imms,
                      std::vector<FEmit> *emits, std::vector<std::pair<uint32_t, uint64_t>> *bias, std::string *error) {
    const std::vector<Peep> peep = peephole(instrs, ps, pl, imms, num_imms);
    uint32_t reg[MAX_REGS];
    for (uint32_t r = 0; r < MAX_REGS; r++) reg[r] = ~0u;
    std::vector<std::pair<uint32_t, uint64_t>> acc[4];
    bool acc_used[4] = {};
    unsigned __int128 acc_bound[4] = {0, 0, 0, 0};
    uint32_t k = 0;
    for (uint32_t pc = ps; pc < ps + pl; pc++) {
        const uint16_t op = instrs[4 * pc], dst = instrs[4 * pc + 1] & (MAX_REGS - 1), a = instrs[4 * pc + 2], b = instrs[4 * pc + 3];
        const uint32_t ra = a & (MAX_REGS - 1), rb = b & (MAX_REGS - 1);
        const Peep &pp = peep[pc - ps];
        switch (op) {
            case GP_LOAD_WIRE: reg[dst] = g.make(FNode::WIRE, a); break;
            case GP_LOAD_CONST: reg[dst] = g.make(FNode::CONST, a); break;
            case GP_LOAD_PI: reg[dst] = g.make(FNode::PI, a & 3); break;
            case GP_LOAD_IMM:
                if (a >= num_imms) {
                    *error = "LOAD_IMM index out of range";
                    return false;
                }
                reg[dst] = g.make(FNode::IMM, ~0u, ~0u, imms[a] % glh::P);
                break;
            case GP_ADD:
            case GP_SUB:
            case GP_MUL:
                if (reg[ra] == ~0u || reg[rb] == ~0u) {
                    *error = "register read before any write";
                    return false;
                }
                switch (pp.kind) {
                    case Peep::PLAIN: reg[dst] = g.make(op == GP_ADD ? FNode::ADD : op == GP_SUB ? FNode::SUB : FNode::MUL, reg[ra], reg[rb]); break;
                    case Peep::ADD_SMALL: reg[dst] = g.make(FNode::ADD_SMALL, reg[pp.src], ~0u, pp.k); break;
                    case Peep::SUB_SMALL: reg[dst] = g.make(FNode::SUB_SMALL, reg[pp.src], ~0u, pp.k); break;
                    case Peep::COPY: reg[dst] = reg[pp.src]; break;
                    case Peep::MUL_ADD1: reg[dst] = g.make(FNode::MUL_ADD1, reg[ra], reg[rb]); break;
                    case Peep::SQUARE: reg[dst] = g.make(FNode::MUL, reg[pp.src], reg[pp.src]); break;
                    case Peep::MUL_K: reg[dst] = g.make(FNode::MUL_K, reg[pp.src], ~0u, pp.k); break;
                    case Peep::ADD_K: reg[dst] = g.make(FNode::ADD_K, reg[pp.src], ~0u, pp.k); break;
                    case Peep::SKIP: break;  // t + 2 of a range check: read by nothing that is still generated
                }
                break;
            case GP_EMIT:
                if (reg[ra] == ~0u) {
                    *error = "register read before any write";
                    return false;
                }
                emits->push_back({k, reg[ra]});
                if (pp.bias) bias->push_back({k, pp.bias});
                k++;
                break;
            case GP_ACC: {
                const uint32_t q = instrs[4 * pc + 1] & 3;
                if (reg[ra] == ~0u || b >= num_imms || imms[b] > 0xFFFFFFFFull) {
                    *error = "ACC: register read before any write, or the immediate is missing / not below 2^32";
                    return false;
                }
                acc_bound[q] += (unsigned __int128)imms[b] * 0xFFFFFFFFull;
                if (acc_bound[q] >> 63) {
                    *error = "ACC: the accumulator could reach 2^63 before its ACCR";
                    return false;
                }
                acc[q].push_back({reg[ra], imms[b]});
                acc_used[q] = true;
                break;
            }
            case GP_ACCR: {
                const uint32_t q = a & 3;
                if (!acc_used[q]) {
                    *error = "ACCR of an accumulator nothing was added to";
                    return false;
                }
                FNode n;
                n.kind = FNode::LIN;
                n.terms = acc[q];
                std::sort(n.terms.begin(), n.terms.end());
                reg[dst] = g.intern(n);
                acc[q].clear();
                acc_bound[q] = 0;
                break;
            }
            case GP_MULK:
                if (reg[ra] == ~0u || b >= 96) {
                    *error = "MULK: register read before any write, or shift >= 96";
                    return false;
                }
                reg[dst] = g.make(FNode::MULK, reg[ra], ~0u, b);
                break;
            default: *error = "unknown opcode"; return false;
        }
    }
    return true;
}

// The fused function of a unit, as statements of the kernel's body. `gate_emits[i]`: the EMITs of the unit's i-th gate.
static void fuse_schedule(std::ostringstream &final_out, const FGraph &g, const std::vector<std::vector<FEmit>> &gate_emits, uint32_t num_selectors) {
    std::ostringstream o;
    const size_t n = g.nodes.size();
    struct Cons {
        uint8_t type;  // 0: node x; 1: EMIT of gate x, constraint y; 2: term y of LIN node x
        uint32_t x, y;
    };
    // only what some EMIT depends on is generated
    std::vector<char> needed(n, 0);
    {
        std::vector<uint32_t> st;
        for (const auto &ge : gate_emits)
            for (const FEmit &e : ge) st.push_back(e.node);
        while (!st.empty()) {
            const uint32_t v = st.back();
            st.pop_back();
            if (needed[v]) continue;
            needed[v] = 1;
            const FNode &nd = g.nodes[v];
            if (nd.a != ~0u && nd.kind != FNode::WIRE && nd.kind != FNode::CONST && nd.kind != FNode::PI) st.push_back(nd.a);
            if (nd.b != ~0u) st.push_back(nd.b);
            for (const auto &t : nd.terms) st.push_back(t.first);
        }
    }
    std::vector<std::vector<Cons>> cons(n);
    std::vector<uint32_t> pending(n, 0);
    for (uint32_t v = 0; v < n; v++) {
        if (!needed[v]) continue;
        const FNode &nd = g.nodes[v];
        const bool leaf = nd.kind == FNode::WIRE || nd.kind == FNode::CONST || nd.kind == FNode::PI || nd.kind == FNode::IMM;
        if (leaf) continue;
        if (nd.kind == FNode::LIN) {
            for (uint32_t t = 0; t < nd.terms.size(); t++) cons[nd.terms[t].first].push_back({2, v, t});
            pending[v] = (uint32_t)nd.terms.size();
            continue;
        }
        cons[nd.a].push_back({0, v, 0});
        pending[v] = 1;
        if (nd.b != ~0u && nd.b != nd.a) {
            cons[nd.b].push_back({0, v, 0});
            pending[v] = 2;
        }
    }
    for (uint32_t gi = 0; gi < gate_emits.size(); gi++)
        for (const FEmit &e : gate_emits[gi]) cons[e.node].push_back({1, gi, e.k});
    // A sum whose terms are computed far apart would keep its accumulator open all the way (four registers): only sums whose terms
    // are numbered — i.e. first computed — close together take their terms as they come; the others are gathered where the first
    // program that needs them stands, re-loading the wires among their terms.
    std::vector<char> gathered(n, 0);
    for (uint32_t v = 0; v < n; v++) {
        if (!needed[v] || g.nodes[v].kind != FNode::LIN) continue;
        uint32_t lo = ~0u, hi = 0;
        for (const auto &t : g.nodes[v].terms) lo = std::min(lo, t.first), hi = std::max(hi, t.first);
        gathered[v] = hi - lo > 400;
    }
    std::vector<char> done(n, 0), lin_open(n, 0);
    std::vector<std::string> name(n);
    std::vector<uint64_t> at(n, 0);
    uint64_t stmt = 0, serial = 0;
    const uint64_t WINDOW = 96;  // statements a loaded wire is kept for; a later reader loads it again
    auto is_load = [&](uint32_t v) { return g.nodes[v].kind == FNode::WIRE || g.nodes[v].kind == FNode::CONST; };
    // `name = <load of v>`: the element stride goes through an asm that returns it unchanged — otherwise the compiler keeps the
    // scalar product index * stride of every wire it has seen (two scalar registers each, 234 wires) for the next load of the same
    // wire, runs out of scalar registers and parks them in lanes of vector registers (600 spills, ten vector registers, in a
    // four-gate unit); a product per load is two scalar instructions. A re-load also hides the pointer (see use()).
    auto load = [&](const std::string &nm, uint32_t v, bool again) {
        const bool wire = g.nodes[v].kind == FNode::WIRE;
        std::ostringstream e;
        // (every such asm carries a number of its own in a comment: identical asm statements of identical inputs are merged)
        e << "{ uint64_t e = " << (wire ? "wes" : "ces") << "; asm(\"; " << ++serial << "\" : \"+s\"(e)); ";
        if (again) e << "const uint64_t* q = " << (wire ? "W" : "C") << "; asm(\"; " << ++serial << "\" : \"+v\"(q)); ";
        e << nm << " = " << (again ? "q" : wire ? "W" : "C") << "[" << (wire ? g.nodes[v].a : num_selectors + g.nodes[v].a) << " * e]; }";
        return e.str();
    };
    // the name to read value v by; a wire loaded long ago is loaded again through a pointer the compiler cannot tell from W (else it
    // would merge the two loads and keep the first one's registers occupied in between)
    auto use = [&](uint32_t v) -> const std::string & {
        if (is_load(v) && stmt - at[v] > WINDOW) {
            std::ostringstream nm;
            nm << "v" << v << "_" << ++serial;
            o << "  uint64_t " << nm.str() << "; " << load(nm.str(), v, true) << "\n";
            name[v] = nm.str();
            at[v] = ++stmt;
        }
        return name[v];
    };
    auto define = [&](uint32_t v) {
        const FNode &nd = g.nodes[v];
        std::ostringstream e;
        switch (nd.kind) {
            case FNode::WIRE: case FNode::CONST:
                name[v] = "v" + std::to_string(v);
                o << "  uint64_t " << name[v] << "; " << load(name[v], v, false) << "\n";
                at[v] = ++stmt;
                return;
            case FNode::PI: e << "g_pih[" << nd.a << "]"; break;
            case FNode::IMM: e << "0x" << std::hex << nd.k << std::dec << "ull"; break;
            case FNode::ADD: { const std::string x = use(nd.a), y = use(nd.b); e << "gl::add(" << x << ", " << y << ")"; break; }
            case FNode::SUB: { const std::string x = use(nd.a), y = use(nd.b); e << "gl::sub(" << x << ", " << y << ")"; break; }
            case FNode::MUL: { const std::string x = use(nd.a), y = use(nd.b); e << "gl::mul(" << x << ", " << y << ")"; break; }
            case FNode::MUL_ADD1: { const std::string x = use(nd.a), y = use(nd.b); e << "gl::mul_add_small<1>(" << x << ", " << y << ")"; break; }
            case FNode::MULK: e << "gl::mul_pow2<" << nd.k << ">(" << use(nd.a) << ")"; break;
            case FNode::ADD_SMALL: e << "gl::add_small<" << nd.k << "u>(" << use(nd.a) << ")"; break;
            case FNode::SUB_SMALL: e << "gl::sub_small<" << nd.k << "u>(" << use(nd.a) << ")"; break;
            case FNode::MUL_K: e << "gl::mul_k<0x" << std::hex << nd.k << std::dec << "ull>(" << use(nd.a) << ")"; break;
            case FNode::ADD_K: e << "gl::add_k<0x" << std::hex << nd.k << std::dec << "ull>(" << use(nd.a) << ")"; break;
            case FNode::LIN:
                if (gathered[v]) {
                    o << "  uint64_t l" << v << "l = 0, l" << v << "h = 0;\n";
                    for (const auto &t : nd.terms) {
                        const std::string x = use(t.first);
                        o << "  gj_acc(l" << v << "l, l" << v << "h, " << x << ", " << t.second << "u);\n";
                        stmt++;
                    }
                }
                e << "gl::fold96(l" << v << "l, l" << v << "h)";
                break;
        }
        name[v] = "v" + std::to_string(v);
        o << "  const uint64_t " << name[v] << " = " << e.str() << ";\n";
        at[v] = ++stmt;
    };
    // Would computing m now lead to an EMIT or to a term of a sum that takes its terms as they come — directly, or through operations
    // that wait for nothing else? If not, m waits until the program that wants it is generated: it would only occupy registers.
    uint32_t budget = 0;  // nodes one question may visit (a wide graph of waiting operations must not make the generator quadratic)
    std::function<bool(uint32_t, int)> fires_from = [&](uint32_t m, int depth) -> bool {
        for (const Cons &c : cons[m]) {
            if (c.type == 1) return true;
            if (c.type == 2 && !gathered[c.x]) return true;
            if (c.type == 0 && pending[c.x] == 1 && depth < 48 && budget > 0 && (--budget, fires_from(c.x, depth + 1))) return true;
        }
        return false;
    };
    auto fires = [&](uint32_t m, int) {
        budget = 512;
        return fires_from(m, 0);
    };
    std::vector<uint32_t> work;
    // compute v (its operands exist), hand it to its consumers, and go on with whatever that completes: depth first, so that a
    // chain is followed to its EMIT before the next value is loaded
    auto produce = [&](uint32_t first) {
        work.push_back(first);
        while (!work.empty()) {
            const uint32_t v = work.back();
            work.pop_back();
            if (done[v]) continue;
            done[v] = 1;
            define(v);
            const size_t mark = work.size();
            for (const Cons &c : cons[v]) {
                if (c.type == 1) {
                    // (the table pointer goes through an asm as well: the gates of a unit number their constraints from zero each, and
                    // the compiler would keep alpha^k in scalar registers from one gate's EMIT to the next gate's with the same k)
                    o << "  { apow_t ap = (apow_t)g_apow; asm(\"; " << ++serial << "\" : \"+s\"(ap)); for (int c = 0; c < NCH; c++) gl::dot_term2(ga" << c.x << "[c], "
                      << name[v] << ", ap[(c * NGC + " << c.y << ") * 2], ap[(c * NGC + " << c.y << ") * 2 + 1]); }\n";
                    stmt++;
                } else if (c.type == 2) {
                    if (gathered[c.x]) continue;
                    if (!lin_open[c.x]) {
                        o << "  uint64_t l" << c.x << "l = 0, l" << c.x << "h = 0;\n";
                        lin_open[c.x] = 1;
                    }
                    o << "  gj_acc(l" << c.x << "l, l" << c.x << "h, " << name[v] << ", " << g.nodes[c.x].terms[c.y].second << "u);\n";
                    stmt++;
                    if (--pending[c.x] == 0) work.push_back(c.x);
                } else if (--pending[c.x] == 0 && fires(c.x, 0)) {
                    work.push_back(c.x);
                }
            }
            std::reverse(work.begin() + (long)mark, work.end());  // first consumer first
        }
    };
    // the order of the gates' programs: nodes are numbered as the programs first reach them, operands before their operations
    for (uint32_t v = 0; v < n; v++)
        if (needed[v] && !done[v]) produce(v);
    // Every load stands where its value is first used, and a wave that waits for each of its thousand loads in turn is not hidden by
    // the three others of its SIMD: the loads move up by `ahead` statements (their order kept), that many statements' work — a few
    // hundred instructions — between the request and the use.
    const uint32_t ahead = fuse_prefetch_distance();
    std::vector<std::string> lines;
    {
        const std::string text = o.str();
        size_t from = 0;
        for (size_t nl; (nl = text.find('\n', from)) != std::string::npos; from = nl + 1) lines.push_back(text.substr(from, nl - from));
    }
    std::vector<std::string> placed;
    size_t last_load = 0;  // no load is placed above the load before it
    for (const std::string &ln : lines) {
        const bool is_a_load = ln.compare(0, 12, "  uint64_t v") == 0 && ln.find(" * e]; }") != std::string::npos;
        if (!is_a_load || ahead == 0) {
            placed.push_back(ln);
            continue;
        }
        // (the barrier after it: left alone, the compiler's scheduler moves the load back down to its use to save the two registers)
        const size_t where = std::max(last_load, placed.size() > ahead ? placed.size() - ahead : 0);
        placed.insert(placed.begin() + (long)where, ln + " __builtin_amdgcn_sched_barrier(0);");
        last_load = where + 1;
    }
    for (const std::string &ln : placed) final_out << ln << "\n";
}

static std::string generate_fused_source(const uint16_t *instrs, uint32_t num_instrs, const uint32_t *gates, const std::vector<uint32_t> &unit_gates,
                                         const uint64_t *imms, uint32_t num_imms, uint32_t num_selectors, uint32_t ngc, uint32_t nch,
                                         std::vector<std::vector<std::pair<uint32_t, uint64_t>>> *bias, std::string *error) {
    std::ostringstream o;
    o << "#define GL_JIT 1\n" << GL_FIELD_SRC << "\n" << GL_JIT_FIELD_SRC << "\n";
    o << "#define NGU " << unit_gates.size() << "\n#define NCH " << nch << "\n#define NGC " << ngc << "\n";
    // g_apow[c][k] = {alpha_c^k, alpha_c^k * 2^32}: the two-column accumulators of gate_jit_field.h (gl::DotCol2)
    // ONE constant object per unit — LDE pointers and strides, public-inputs hash, alpha powers, biases — so that a launch uploads
    // one piece per unit (round 5: four pieces per unit, 24 small copies per quotient); the names the generated code uses are macros
    o << "struct GateTab { uint64_t par[6]; uint64_t pih[4]; uint64_t apow[NCH * NGC * 2]; uint64_t bias[NCH * NGU]; };\n__constant__ GateTab g_tab;\n"
         "#define g_apow g_tab.apow\n#define g_pih g_tab.pih\n#define g_bias g_tab.bias\n#define g_par g_tab.par\n"
         "typedef const __attribute__((address_space(4))) uint64_t* apow_t;\n"
         "static __device__ __forceinline__ void gj_acc(uint64_t &al, uint64_t &ah, uint64_t x, uint32_t k) {\n"
         "  asm(\"v_mad_u64_u32 %0, vcc, %2, %4, %0\\n\\tv_mad_u64_u32 %1, vcc, %3, %4, %1\" : \"+v\"(al), \"+v\"(ah) : \"v\"((uint32_t)x), \"v\"((uint32_t)(x >> 32)), \"s\"(k) : \"vcc\");\n}\n";
    bias->assign(unit_gates.size(), {});
    FGraph graph;
    std::vector<std::vector<FEmit>> gate_emits(unit_gates.size());
    for (size_t gi = 0; gi < unit_gates.size(); gi++) {
        const uint32_t *d = gates + 6 * unit_gates[gi];
        const uint32_t si = d[1], gs = d[2], ge = d[3], ps = d[4], pl = d[5];
        if (ps + pl > num_instrs || si >= num_selectors || gs > ge) {
            *error = "gate descriptor out of range";
            return "";
        }
        if (!fuse_gate(graph, instrs, ps, pl, imms, num_imms, &gate_emits[gi], &(*bias)[gi], error)) return "";
    }
    o << "extern \"C\" __global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(" << fuse_waves() << ", " << fuse_waves()
      << "))) void gate_constraints_kernel(const uint64_t* __restrict__ wires, uint64_t wrs, "
         "uint64_t wes_, const uint64_t* __restrict__ cs, uint64_t crs, uint64_t ces_, uint64_t lde_size, uint64_t* __restrict__ out, int accumulate) {\n"
         "  const uint64_t t_ = (uint64_t)blockIdx.x * 128u + threadIdx.x;\n  if (t_ >= lde_size) return;\n"
         "  const uint64_t* W = (const uint64_t*)g_par[0] + t_ * g_par[1]; const uint64_t wes = g_par[2];\n"
         "  const uint64_t* C = (const uint64_t*)g_par[3] + t_ * g_par[4]; const uint64_t ces = g_par[5];\n";
    for (size_t gi = 0; gi < unit_gates.size(); gi++)
        if (!gate_emits[gi].empty()) o << "  gl::DotCol2 ga" << gi << "[NCH];  // gate_" << unit_gates[gi] << "\n";
    fuse_schedule(o, graph, gate_emits, num_selectors);
    o << "  uint64_t acc[NCH];\n  for (int c = 0; c < NCH; c++) acc[c] = accumulate ? out[(uint64_t)c * lde_size + t_] : 0;\n";
    for (size_t gi = 0; gi < unit_gates.size(); gi++) {
        if (gate_emits[gi].empty()) continue;  // a gate without constraints (NoopGate) contributes nothing
        const uint32_t *d = gates + 6 * unit_gates[gi];
        const uint32_t row = d[0], si = d[1], gs = d[2], ge = d[3];
        // compute_filter (gates/gate.rs:261-268)
        o << "  {\n    const uint64_t s = C[" << si << " * ces];\n    uint64_t filt = 1;\n";
        for (uint32_t i = gs; i < ge; i++)
            if (i != row) o << "    filt = gl::mul(filt, gl::sub(" << i << "ull, s));\n";
        if (num_selectors > 1) o << "    filt = gl::mul(filt, gl::sub(0xFFFFFFFFull, s));\n";  // UNUSED_SELECTOR (selectors.rs:11)
        if ((*bias)[gi].empty())
            o << "    for (int c = 0; c < NCH; c++) acc[c] = gl::add(acc[c], gl::mul(filt, gl::dot_finish2(ga" << gi << "[c])));\n  }\n";
        else
            o << "    for (int c = 0; c < NCH; c++) acc[c] = gl::add(acc[c], gl::mul(filt, gl::sub(gl::dot_finish2(ga" << gi << "[c]), g_bias[c * NGU + "
              << gi << "])));\n  }\n";
    }
    o << "  for (int c = 0; c < NCH; c++) out[(uint64_t)c * lde_size + t_] = gl::canon(acc[c]);\n}\n";
    return o.str();
}

// Source of one unit: the device functions of `unit_gates` and the kernel that calls them. `bias`: per gate of the unit, the
// constraints emitted with a constant added.
static std::string generate_source(const uint16_t *instrs, uint32_t num_instrs, const uint32_t *gates, const std::vector<uint32_t> &unit_gates,
                                   const uint64_t *imms, uint32_t num_imms, uint32_t num_selectors, uint32_t ngc, uint32_t nch,
                                   std::vector<std::vector<std::pair<uint32_t, uint64_t>>> *bias, std::string *error) {
    if (fuse_enabled()) return generate_fused_source(instrs, num_instrs, gates, unit_gates, imms, num_imms, num_selectors, ngc, nch, bias, error);
    std::ostringstream o;
    o << "#define GL_JIT 1\n" << GL_FIELD_SRC << "\n" << GL_JIT_FIELD_SRC << "\n";
    o << "#define NGU " << unit_gates.size() << "\n";
    bias->assign(unit_gates.size(), {});
    o << "#define NCH " << nch << "\n#define NGC " << ngc << "\n";
    o << "struct GateSum { uint64_t v[NCH]; };\n";
    // alpha powers and the public-inputs hash live at link-time-constant addresses, so every read is a scalar
    // load (a pointer ARGUMENT of a non-inlined device function arrives in VGPRs and would be read per lane)
    o << "struct GateTab { uint64_t par[6]; uint64_t pih[4]; uint64_t apow[NCH * NGC]; uint64_t bias[NCH * NGU]; };\n__constant__ GateTab g_tab;\n"
         "#define g_apow g_tab.apow\n#define g_pih g_tab.pih\n#define g_bias g_tab.bias\n#define g_par g_tab.par\n";
    // Where the LDE lives: {wires, row stride, element stride, constants/sigmas, row stride, element stride} (strides in elements), as
    // link-time-constant scalars for the same reason: the element stride that every wire load multiplies by is then a scalar, not a
    // vector register of a function argument. (Measured in round 5, profiles/r05_quotient_codegen_ab.jsonl: going further — a BUFFER
    // load per wire whose descriptor carries the uniform part of the address, zero vector instructions per load, 16 k of 165 k fewer —
    // made the ed25519 quotient 2 % SLOWER: the scalar chain that rebuilds the descriptor in front of every load keeps the compiler from
    // issuing a gate's loads in one batch, and four waves per SIMD do not hide the latency that exposes.)
    o << "static __device__ __forceinline__ void gj_acc(uint64_t &al, uint64_t &ah, uint64_t x, uint32_t k) {\n"
         "  asm(\"v_mad_u64_u32 %0, vcc, %2, %4, %0\\n\\tv_mad_u64_u32 %1, vcc, %3, %4, %1\" : \"+v\"(al), \"+v\"(ah) : \"v\"((uint32_t)x), \"v\"((uint32_t)(x >> 32)), \"s\"(k) : \"vcc\");\n}\n";
    for (size_t gi = 0; gi < unit_gates.size(); gi++) {
        const uint32_t g = unit_gates[gi];
        const uint32_t *d = gates + 6 * g;
        const uint32_t row = d[0], si = d[1], gs = d[2], ge = d[3], ps = d[4], pl = d[5];
        if (ps + pl > num_instrs || si >= num_selectors || gs > ge) {
            *error = "gate descriptor out of range";
            return "";
        }
        const std::vector<Peep> peep = peephole(instrs, ps, pl, imms, num_imms);
        o << "static __device__ __noinline__ GateSum gate_" << g << "() {\n"
             "  const uint64_t t_ = (uint64_t)blockIdx.x * 128u + threadIdx.x;\n"   // the kernel runs 128 lanes per block
             "  const uint64_t* W = (const uint64_t*)g_par[0] + t_ * g_par[1]; const uint64_t wes = g_par[2];\n"
             "  const uint64_t* C = (const uint64_t*)g_par[3] + t_ * g_par[4]; const uint64_t ces = g_par[5];\n";
        // compute_filter (gates/gate.rs:261-268)
        o << "  const uint64_t s = C[" << si << " * ces];\n  uint64_t filt = 1;\n";
        for (uint32_t i = gs; i < ge; i++)
            if (i != row) o << "  filt = gl::mul(filt, gl::sub(" << i << "ull, s));\n";
        if (num_selectors > 1) o << "  filt = gl::mul(filt, gl::sub(0xFFFFFFFFull, s));\n";  // UNUSED_SELECTOR (selectors.rs:11)
        // The gate's constraints are reduced with powers of alpha lazily: one 192-bit column accumulator per challenge, one
        // reduction per gate (gl::DotAcc: four multiply-adds and four carry counts per term) instead of a multiply-reduce-add per
        // constraint. (Tried in round 3: the powers in 22-bit limbs, six multiply-adds per term and no carry counts — 5 % fewer
        // vector instructions and 2.5 % SLOWER: a v_mad_u64_u32 costs about two plain instructions here. LABNOTES.md 10.)
        o << "  gl::DotAcc ga[NCH];\n";
        bool used[MAX_REGS] = {};
        bool acc_used[4] = {};
        for (uint32_t pc = ps; pc < ps + pl; pc++) {
            if (instrs[4 * pc] == GP_ACC)  // its dst field names an accumulator, not a register
                acc_used[instrs[4 * pc + 1] & 3] = true;
            else
                used[instrs[4 * pc + 1] & (MAX_REGS - 1)] = true;
        }
        for (int q = 0; q < 4; q++)
            if (acc_used[q]) o << "  uint64_t acc" << q << "l = 0, acc" << q << "h = 0;\n";
        // worst case of each accumulator half: sum of imm * (2^32 - 1) since its last ACCR; gl::fold96 needs < 2^63
        unsigned __int128 acc_bound[4] = {0, 0, 0, 0};
        for (uint32_t r = 0; r < MAX_REGS; r++)
            if (used[r]) o << "  uint64_t r" << r << " = 0;\n";
        uint32_t k = 0;
        for (uint32_t pc = ps; pc < ps + pl; pc++) {
            const uint16_t op = instrs[4 * pc], dst = instrs[4 * pc + 1] & (MAX_REGS - 1), a = instrs[4 * pc + 2], b = instrs[4 * pc + 3];
            const uint32_t ra = a & (MAX_REGS - 1), rb = b & (MAX_REGS - 1);
            const Peep &pp = peep[pc - ps];
            if (pp.kind != Peep::PLAIN && (op == GP_ADD || op == GP_SUB || op == GP_MUL)) {
                if (!used[ra] || !used[rb]) {
                    *error = "register read before any write";
                    return "";
                }
                switch (pp.kind) {
                    case Peep::ADD_SMALL: o << "  r" << dst << " = gl::add_small<" << pp.k << "u>(r" << pp.src << ");\n"; break;
                    case Peep::SUB_SMALL: o << "  r" << dst << " = gl::sub_small<" << pp.k << "u>(r" << pp.src << ");\n"; break;
                    case Peep::COPY: o << "  r" << dst << " = r" << pp.src << ";\n"; break;
                    case Peep::MUL_ADD1: o << "  r" << dst << " = gl::mul_add_small<1>(r" << ra << ", r" << rb << ");\n"; break;
                    case Peep::SQUARE: o << "  r" << dst << " = gl::mul(r" << pp.src << ", r" << pp.src << ");\n"; break;
                    case Peep::MUL_K: o << "  r" << dst << " = gl::mul_k<0x" << std::hex << pp.k << std::dec << "ull>(r" << pp.src << ");\n"; break;
                    case Peep::ADD_K: o << "  r" << dst << " = gl::add_k<0x" << std::hex << pp.k << std::dec << "ull>(r" << pp.src << ");\n"; break;
                    default: break;  // SKIP: the value is read by nothing the generated code still contains
                }
                continue;
            }
            if (pp.kind == Peep::SKIP && op == GP_LOAD_IMM) continue;  // every reader takes the constant as an operand
            switch (op) {
                case GP_LOAD_WIRE: o << "  r" << dst << " = W[" << a << " * wes];\n"; break;
                case GP_LOAD_CONST: o << "  r" << dst << " = C[" << (num_selectors + a) << " * ces];\n"; break;
                case GP_LOAD_PI: o << "  r" << dst << " = g_pih[" << (a & 3) << "];\n"; break;
                case GP_LOAD_IMM:
                    if (a >= num_imms) {
                        *error = "LOAD_IMM index out of range";
                        return "";
                    }
                    o << "  r" << dst << " = 0x" << std::hex << (imms[a] % glh::P) << std::dec << "ull;\n";
                    break;
                case GP_ADD:
                case GP_SUB:
                case GP_MUL:
                    if (!used[ra] || !used[rb]) {
                        *error = "register read before any write";
                        return "";
                    }
                    o << "  r" << dst << " = gl::" << (op == GP_ADD ? "add" : op == GP_SUB ? "sub" : "mul") << "(r" << ra << ", r" << rb
                      << ");\n";
                    break;
                case GP_EMIT:
                    if (!used[ra]) {
                        *error = "register read before any write";
                        return "";
                    }
                    o << "  for (int c = 0; c < NCH; c++) gl::dot_term(ga[c], r" << ra << ", g_apow[c * NGC + " << k << "]);\n";  // k < ngc: validated
                    if (pp.bias) (*bias)[gi].push_back({k, pp.bias});
                    k++;
                    break;
                case GP_ACC: {  // acc[dst] += r[a] * imm[b]: two 32x32+64 multiply-adds, no modular step
                    const uint32_t q = instrs[4 * pc + 1] & 3;
                    if (!used[ra] || b >= num_imms || imms[b] > 0xFFFFFFFFull) {
                        *error = "ACC: register read before any write, or the immediate is missing / not below 2^32";
                        return "";
                    }
                    acc_bound[q] += (unsigned __int128)imms[b] * 0xFFFFFFFFull;
                    if (acc_bound[q] >> 63) {
                        *error = "ACC: the accumulator could reach 2^63 before its ACCR";
                        return "";
                    }
                    // both halves as ONE statement of two multiply-adds: left to itself the compiler strength-reduces the small constant
                    // weights (x 4^j: a 64-bit shift, four ANDs and two 64-bit adds, 24 issue cycles where two multiply-adds take 12)
                    o << "  gj_acc(acc" << q << "l, acc" << q << "h, r" << ra << ", " << imms[b] << "u);\n";
                    break;
                }
                case GP_ACCR: {  // r[dst] = acc[a] mod p; acc[a] = 0
                    const uint32_t q = a & 3;
                    if (!acc_used[q]) {
                        *error = "ACCR of an accumulator nothing was added to";
                        return "";
                    }
                    o << "  r" << dst << " = gl::fold96(acc" << q << "l, acc" << q << "h); acc" << q << "l = 0; acc" << q << "h = 0;\n";
                    acc_bound[q] = 0;
                    break;
                }
                case GP_MULK:
                    if (!used[ra] || b >= 96) {
                        *error = "MULK: register read before any write, or shift >= 96";
                        return "";
                    }
                    o << "  r" << dst << " = gl::mul_pow2<" << b << ">(r" << ra << ");\n";
                    break;
                default: *error = "unknown opcode"; return "";
            }
        }
        if ((*bias)[gi].empty())
            o << "  GateSum out;\n  for (int c = 0; c < NCH; c++) out.v[c] = gl::mul(filt, gl::dot_finish(ga[c]));\n  return out;\n}\n";
        else
            o << "  GateSum out;\n  for (int c = 0; c < NCH; c++) out.v[c] = gl::mul(filt, gl::sub(gl::dot_finish(ga[c]), g_bias[c * NGU + " << gi
              << "]));\n  return out;\n}\n";
    }
    // `accumulate`: the output already holds the sum of the units that ran before this one
    o << "extern \"C\" __global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(4, 4))) void gate_constraints_kernel(const uint64_t* __restrict__ wires, uint64_t wrs, "
         "uint64_t wes, const uint64_t* __restrict__ cs, uint64_t crs, uint64_t ces, uint64_t lde_size, uint64_t* __restrict__ out, int accumulate) {\n"
         "  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;\n  if (t >= lde_size) return;\n"
         "  uint64_t acc[NCH];\n  for (int c = 0; c < NCH; c++) acc[c] = accumulate ? out[(uint64_t)c * lde_size + t] : 0;\n";
    for (uint32_t g : unit_gates)
        o << "  { GateSum s = gate_" << g << "(); for (int c = 0; c < NCH; c++) acc[c] = gl::add(acc[c], s.v[c]); }\n";
    o << "  for (int c = 0; c < NCH; c++) out[(uint64_t)c * lde_size + t] = gl::canon(acc[c]);\n}\n";
    return o.str();
}

// How many units a circuit is cut into: PLONKY2_HIP_JIT_UNITS (1 = one program, as before round 3), else one per hardware
// thread up to eight.
static uint32_t jit_unit_limit() {
    if (const char *e = PLONKY2_KNOB("PLONKY2_HIP_JIT_UNITS")) {
        const long v = strtol(e, nullptr, 10);
        if (v >= 1 && v <= 64) return (uint32_t)v;
    }
    const unsigned hw = std::thread::hardware_concurrency();
    return hw == 0 ? 4 : hw > 8 ? 8 : hw;
}

// Cache file name of a unit (kernel_cache_dir()): keyed by a hash of the generated source and of what turns the same source
// into a different code object — the hiprtc version and the target.
static std::string unit_cache_path(const std::string &dir, const std::string &source) {
    if (dir.empty()) return "";
    int rtc_major = 0, rtc_minor = 0;
    (void)hiprtcVersion(&rtc_major, &rtc_minor);
    const std::string salt = "hiprtc " + std::to_string(rtc_major) + "." + std::to_string(rtc_minor) + " " + JIT_ARCH + "\n";
    uint64_t h = 0xcbf29ce484222325ull;  // FNV-1a
    for (unsigned char ch : salt) h = (h ^ ch) * 0x100000001b3ull;
    for (unsigned char ch : source) h = (h ^ ch) * 0x100000001b3ull;
    char name[64];
    snprintf(name, sizeof name, "/gate_%016llx", (unsigned long long)h);
    return dir + name;
}

// hiprtc on one unit's source; the code object (and the source, for inspection) goes to the cache. Runs on a thread of its own.
static bool compile_unit(GateUnit &u, const std::string &cache_path) {
    hiprtcProgram prog;
    hiprtcResult r = hiprtcCreateProgram(&prog, u.source.c_str(), "gate_constraints.hip", 0, nullptr, nullptr);
    if (r != HIPRTC_SUCCESS) {
        u.error = std::string("hiprtcCreateProgram: ") + hiprtcGetErrorString(r);
        return false;
    }
    const std::string arch = std::string("--offload-arch=") + JIT_ARCH;
    const char *opts[] = {arch.c_str(), "-O3", "-std=c++17"};
    r = hiprtcCompileProgram(prog, 3, opts);
    if (r != HIPRTC_SUCCESS) {
        size_t ls = 0;
        hiprtcGetProgramLogSize(prog, &ls);
        std::string log(ls, '\0');
        if (ls) hiprtcGetProgramLog(prog, &log[0]);
        u.error = std::string("hiprtcCompileProgram: ") + hiprtcGetErrorString(r) + "\n" + log.substr(0, 4000);
        hiprtcDestroyProgram(&prog);
        return false;
    }
    size_t cs = 0;
    hiprtcGetCodeSize(prog, &cs);
    u.code.resize(cs);
    hiprtcGetCode(prog, u.code.data());
    hiprtcDestroyProgram(&prog);
    if (!cache_path.empty()) {
        // Several processes (one per GPU) may build the same circuit at once: each writes a file of its own and
        // renames it into place, so a reader sees either nothing or a whole code object.
        const std::string pid = std::to_string((long long)getpid()) + "." + std::to_string((unsigned long long)(uintptr_t)&u);
        const std::string tmp = cache_path + ".tmp." + pid, tmp_src = cache_path + ".hip." + pid;
        std::ofstream(tmp_src) << u.source;
        (void)rename(tmp_src.c_str(), (cache_path + ".hip").c_str());
        bool written = false;
        {
            std::ofstream f(tmp, std::ios::binary);
            f.write(u.code.data(), (std::streamsize)u.code.size());
            f.flush();
            written = f.good();
        }
        if (!written || rename(tmp.c_str(), (cache_path + ".hsaco").c_str()) != 0) (void)remove(tmp.c_str());
    }
    return true;
}

static hipError_t load_unit(GateUnit &u) {
    hipError_t e = hipModuleLoadData(&u.module, u.code.data());
    if (e == hipSuccess) e = hipModuleGetFunction(&u.fn, u.module, "gate_constraints_kernel");
    size_t bytes = 0;
    if (e == hipSuccess) e = hipModuleGetGlobal(reinterpret_cast<hipDeviceptr_t *>(&u.d_tab), &bytes, u.module, "g_tab");
    u.biased = false;
    for (const auto &b : u.bias) u.biased |= !b.empty();
    return e;
}

// Which gates share a fused unit: a unit is opened by the first gate (in circuit order) that has none, and takes in, one at a time,
// the gate that would recompute least — the largest share of its own operations (priced as vector instructions) already in the
// unit's value graph — while that share is above a quarter, the unit has fewer than fuse_gates_per_unit() gates and its value graph
// stays below 4 500 nodes (hiprtc's time grows faster than the function). Gates without constraints belong to no unit.
static bool fuse_partition(const uint16_t *instrs, uint32_t num_instrs, const uint32_t *gates, uint32_t num_gates, const uint64_t *imms,
                           uint32_t num_imms, uint32_t num_selectors, std::vector<GateUnit> *units, std::string *error) {
    FGraph graph;
    std::vector<std::vector<uint32_t>> nodes_of(num_gates);
    for (uint32_t g = 0; g < num_gates; g++) {
        const uint32_t *d = gates + 6 * g;
        if ((uint64_t)d[4] + d[5] > num_instrs || d[1] >= num_selectors || d[2] > d[3]) {
            *error = "gate descriptor out of range";
            return false;
        }
        std::vector<FEmit> emits;
        std::vector<std::pair<uint32_t, uint64_t>> bias;
        if (!fuse_gate(graph, instrs, d[4], d[5], imms, num_imms, &emits, &bias, error)) return false;
        std::vector<uint32_t> st;
        for (const FEmit &e : emits) st.push_back(e.node);
        std::vector<uint32_t> &seen = nodes_of[g];
        std::vector<char> mark(graph.nodes.size(), 0);
        while (!st.empty()) {
            const uint32_t v = st.back();
            st.pop_back();
            if (mark[v]) continue;
            mark[v] = 1;
            seen.push_back(v);
            const FNode &nd = graph.nodes[v];
            if (nd.a != ~0u && nd.kind != FNode::WIRE && nd.kind != FNode::CONST && nd.kind != FNode::PI) st.push_back(nd.a);
            if (nd.b != ~0u) st.push_back(nd.b);
            for (const auto &t : nd.terms) st.push_back(t.first);
        }
    }
    auto price = [&](uint32_t v) -> uint64_t {
        const FNode &nd = graph.nodes[v];
        switch (nd.kind) {
            case FNode::MUL: case FNode::MUL_ADD1: case FNode::MUL_K: return 12;
            case FNode::ADD: case FNode::ADD_K: return 4;
            case FNode::SUB: return 5;
            case FNode::MULK: return 8;
            case FNode::LIN: return 7 + 2 * nd.terms.size();
            case FNode::IMM: case FNode::PI: return 0;
            default: return 2;
        }
    };
    const uint32_t per = fuse_gates_per_unit();
    std::vector<char> assigned(num_gates, 0), in_unit(graph.nodes.size(), 0);
    std::vector<uint64_t> unit_size;
    for (uint32_t g = 0; g < num_gates; g++) {
        if (assigned[g] || nodes_of[g].empty()) continue;
        GateUnit u;
        std::fill(in_unit.begin(), in_unit.end(), 0);
        uint64_t unit_nodes = 0;
        uint32_t next = g;
        while (true) {
            assigned[next] = 1;
            u.gates.push_back(next);
            for (uint32_t v : nodes_of[next]) unit_nodes += !in_unit[v], in_unit[v] = 1;
            if (u.gates.size() >= per) break;
            double best = 0.25;
            next = ~0u;
            for (uint32_t h = g + 1; h < num_gates; h++) {
                if (assigned[h] || nodes_of[h].empty()) continue;
                uint64_t all = 0, shared = 0, fresh = 0;
                for (uint32_t v : nodes_of[h]) all += price(v), shared += in_unit[v] ? price(v) : 0, fresh += !in_unit[v];
                const double share = all ? (double)shared / (double)all : 0.0;
                if (unit_nodes + fresh <= 4500 && share > best) best = share, next = h;
            }
            if (next == ~0u) break;
        }
        std::sort(u.gates.begin(), u.gates.end());
        unit_size.push_back(unit_nodes);
        units->push_back(std::move(u));
    }
    // small units that found no company share a launch all the same (a launch streams the selectors and the output once more)
    for (size_t i = 0; i < units->size(); i++) {
        if (unit_size[i] >= 600) continue;
        for (size_t j = i + 1; j < units->size();) {
            if (unit_size[j] < 600 && unit_size[i] + unit_size[j] < 1200 && (*units)[i].gates.size() + (*units)[j].gates.size() <= per) {
                (*units)[i].gates.insert((*units)[i].gates.end(), (*units)[j].gates.begin(), (*units)[j].gates.end());
                unit_size[i] += unit_size[j];
                units->erase(units->begin() + (long)j);
                unit_size.erase(unit_size.begin() + (long)j);
            } else {
                j++;
            }
        }
        std::sort((*units)[i].gates.begin(), (*units)[i].gates.end());
    }
    if (units->empty()) {  // no gate has a constraint: one unit that writes zeros
        GateUnit u;
        u.gates.push_back(0);
        units->push_back(std::move(u));
    }
    return true;
}

GateKernel *gate_kernel_build(const uint16_t *instrs, uint32_t num_instrs, const uint32_t *gates, uint32_t num_gates,
                              const uint64_t *imms, uint32_t num_imms, uint32_t num_selectors, uint32_t num_gate_constraints,
                              uint32_t num_challenges, std::string *error) {
    if (num_challenges == 0 || num_challenges > MAX_CH || num_gates == 0 || num_gate_constraints == 0) {
        *error = "bad gate kernel shape";
        return nullptr;
    }
    GateKernel *k = new GateKernel();
    k->num_challenges = num_challenges;
    k->num_constraints = num_gate_constraints;
    if (!gate_programs_validate(instrs, num_instrs, gates, num_gates, num_imms, num_selectors, num_gate_constraints, &k->wires_needed,
                                &k->constants_needed, error)) {
        delete k;
        return nullptr;
    }
    k->fused = fuse_enabled();
    if (k->fused) {
        if (!fuse_partition(instrs, num_instrs, gates, num_gates, imms, num_imms, num_selectors, &k->units, error)) {
            delete k;
            return nullptr;
        }
    } else {
        // Units of about equal program length: longest gate first, each into the unit that is shortest so far; inside a unit the
        // gates keep the circuit's order.
        const uint32_t n_units = std::min(jit_unit_limit(), num_gates);
        std::vector<uint32_t> order(num_gates);
        for (uint32_t g = 0; g < num_gates; g++) order[g] = g;
        std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return gates[6 * x + 5] > gates[6 * y + 5]; });
        k->units.resize(n_units);
        std::vector<uint64_t> load(n_units, 0);
        for (uint32_t g : order) {
            const uint32_t u = (uint32_t)(std::min_element(load.begin(), load.end()) - load.begin());
            k->units[u].gates.push_back(g);
            load[u] += gates[6 * g + 5] + 16;  // + the gate's fixed part (filter, reduction)
        }
        k->units.erase(std::remove_if(k->units.begin(), k->units.end(), [](const GateUnit &u) { return u.gates.empty(); }), k->units.end());
    }
    const std::string dir = kernel_cache_dir();
    std::vector<std::string> cache_paths;
    for (GateUnit &u : k->units) {
        std::sort(u.gates.begin(), u.gates.end());
        u.source = generate_source(instrs, num_instrs, gates, u.gates, imms, num_imms, num_selectors, num_gate_constraints, num_challenges, &u.bias,
                                   error);
        if (u.source.empty()) {
            delete k;
            return nullptr;
        }
        k->source += u.source;
        cache_paths.push_back(unit_cache_path(dir, u.source));
    }
    // $PLONKY2_HIP_KERNEL_CACHE_LIST=<file>: the cache entries this build uses, one path per line (appended) — how build() tells the
    // current generator's units from whatever else the cache directory holds without compiling anything twice (__graft_entry__.py)
    if (const char *lf = getenv("PLONKY2_HIP_KERNEL_CACHE_LIST"))
        if (FILE *f = fopen(lf, "a")) {
            for (const std::string &cp : cache_paths)
                if (!cp.empty()) fprintf(f, "%s\n", cp.c_str());
            fclose(f);
        }
    auto from_cache = [&](size_t i) {
        if (cache_paths[i].empty()) return false;
        std::ifstream f(cache_paths[i] + ".hsaco", std::ios::binary);
        if (f) k->units[i].code.assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
        return !k->units[i].code.empty();
    };
    // Compile what the cache does not have. hiprtc serialises concurrent compilations of one process behind a lock of its own
    // (eight threads took the 68 s one thread takes), so every unit but the first goes to a forked child: the child runs
    // hiprtc only — nothing that touches a device — writes the code object to a file and leaves with _exit; the parent compiles
    // the first unit itself meanwhile, then collects the files. Forking is OPT-IN (PLONKY2_HIP_JIT_FORK=1): a library must not
    // fork a host that is multi-threaded and has HIP/ROCr initialised (locks held by other threads stay locked in the child,
    // the child inherits the KFD descriptors), so by default the units are compiled in this process, one after the other, and the
    // on-disk cache is what makes the second call fast. build() (__graft_entry__.py) opts in: it runs before any HIP call, in a
    // single-threaded process. A child that fails falls back to compiling here.
    auto compile_missing = [&](const std::vector<size_t> &which) -> bool {
        const char *fk = getenv("PLONKY2_HIP_JIT_FORK");
        const bool may_fork = fk && fk[0] == '1' && which.size() > 1;
        struct Child {
            pid_t pid;
            size_t unit;
            std::string file;
        };
        std::vector<Child> children;
        if (may_fork) {
            for (size_t j = 1; j < which.size(); j++) {
                const size_t i = which[j];
                char tmpl[] = "/tmp/plonky2_hip_jit_XXXXXX";
                const int fd = mkstemp(tmpl);
                if (fd < 0) break;
                close(fd);
                const pid_t pid = fork();
                if (pid < 0) {
                    (void)remove(tmpl);
                    break;
                }
                if (pid == 0) {
                    GateUnit &u = k->units[i];
                    bool ok = compile_unit(u, cache_paths[i]);
                    if (ok) {
                        std::ofstream f(tmpl, std::ios::binary);
                        f.write(u.code.data(), (std::streamsize)u.code.size());
                        f.flush();
                        ok = f.good();
                    }
                    _exit(ok ? 0 : 1);
                }
                children.push_back(Child{pid, i, tmpl});
            }
        }
        std::vector<char> done(k->units.size(), 0);
        // this process: the first unit, and whatever could not be forked
        for (size_t j = 0; j < which.size(); j++) {
            const size_t i = which[j];
            bool forked = false;
            for (const Child &c : children) forked |= c.unit == i;
            if (forked) continue;
            if (!compile_unit(k->units[i], cache_paths[i])) {
                *error = k->units[i].error;
                for (const Child &c : children) {
                    (void)kill(c.pid, SIGKILL);
                    (void)waitpid(c.pid, nullptr, 0);
                    (void)remove(c.file.c_str());
                }
                return false;
            }
            done[i] = 1;
        }
        const time_t deadline = time(nullptr) + 180;  // a unit of the largest table (ed25519 / 8) compiles in about 10 s
        for (const Child &c : children) {
            int status = 0;
            pid_t w = 0;
            while ((w = waitpid(c.pid, &status, WNOHANG)) == 0 && time(nullptr) < deadline) usleep(20000);
            if (w == 0) {
                (void)kill(c.pid, SIGKILL);
                (void)waitpid(c.pid, &status, 0);
                status = -1;
            }
            if (w >= 0 && WIFEXITED(status) && WEXITSTATUS(status) == 0) {
                std::ifstream f(c.file, std::ios::binary);
                if (f) k->units[c.unit].code.assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
                done[c.unit] = !k->units[c.unit].code.empty();
            }
            (void)remove(c.file.c_str());
        }
        for (const Child &c : children)
            if (!done[c.unit] && !compile_unit(k->units[c.unit], cache_paths[c.unit])) {  // here the error text is available
                *error = k->units[c.unit].error;
                return false;
            }
        return true;
    };
    std::vector<size_t> missing;
    std::vector<char> cached(k->units.size(), 0);
    for (size_t i = 0; i < k->units.size(); i++) {
        cached[i] = from_cache(i) ? 1 : 0;
        if (!cached[i]) missing.push_back(i);
    }
    if (!compile_missing(missing)) {
        delete k;
        return nullptr;
    }
    int ndev = 0;
    const bool have_device = hipGetDeviceCount(&ndev) == hipSuccess && ndev > 0;
    for (size_t i = 0; i < k->units.size(); i++) {
        GateUnit &u = k->units[i];
        hipError_t e = load_unit(u);
        if (e != hipSuccess && cached[i] && have_device) {
            // a cached object that does not load (written by another ROCm, damaged): drop it and compile
            (void)hipGetLastError();
            if (u.module) (void)hipModuleUnload(u.module);
            u.module = nullptr;
            (void)remove((cache_paths[i] + ".hsaco").c_str());
            u.code.clear();
            if (!compile_missing({i})) {
                gate_kernel_destroy(k);
                return nullptr;
            }
            e = load_unit(u);
        }
        if (e != hipSuccess) {
            *error = std::string("loading the compiled gate kernel: ") + hipGetErrorString(e);
            gate_kernel_destroy(k);
            return nullptr;
        }
        u.code.clear();
        u.code.shrink_to_fit();
    }
    return k;
}

void gate_kernel_destroy(GateKernel *k) {
    if (!k) return;
    if (k->launched) (void)hipEventSynchronize(k->done);
    if (k->done) (void)hipEventDestroy(k->done);
    for (GateUnit &u : k->units)
        if (u.module) (void)hipModuleUnload(u.module);
    delete k;
}

uint32_t gate_kernel_num_challenges(const GateKernel *k) { return k->num_challenges; }
uint32_t gate_kernel_num_constraints(const GateKernel *k) { return k->num_constraints; }
uint32_t gate_kernel_wires_needed(const GateKernel *k) { return k->wires_needed; }
uint32_t gate_kernel_constants_needed(const GateKernel *k) { return k->constants_needed; }
const char *gate_kernel_source(const GateKernel *k) { return k->source.c_str(); }

hipError_t gate_kernel_launch(const GateKernel *kc, const uint64_t *wires, uint64_t w_rs, uint64_t w_es, const uint64_t *cs,
                              uint64_t c_rs, uint64_t c_es, const uint64_t *alphas, const uint64_t pih[4], uint64_t lde_size,
                              uint64_t *out, hipStream_t stream) {
    GateKernel *k = const_cast<GateKernel *>(kc);  // the launch state is the object's own (see GateKernel)
    std::lock_guard<std::mutex> turn(k->launch_mu);
    hipError_t e = hipSuccess;
    if (!k->done) {
        e = hipEventCreateWithFlags(&k->done, hipEventDisableTiming);
        if (e != hipSuccess) return e;
    }
    if (k->launched) {
        // the previous launch (normally the previous proof's, long finished): its copies have left the host tables and its kernels
        // have read the constant tables once this event has fired; other streams wait for it on the device, the host only here
        e = hipEventSynchronize(k->done);
        if (e != hipSuccess) return e;
    }
    std::vector<uint64_t> apow((size_t)k->num_challenges * k->num_constraints);
    for (uint32_t c = 0; c < k->num_challenges; c++) {
        uint64_t a = alphas[c] % glh::P, p = 1;
        for (uint32_t j = 0; j < k->num_constraints; j++) {
            apow[(size_t)c * k->num_constraints + j] = p;
            p = glh::mul(p, a);
        }
    }
    std::vector<uint64_t> table = apow;  // what the units' g_apow holds
    if (k->fused) {
        table.resize(apow.size() * 2);
        for (size_t i = 0; i < apow.size(); i++) table[2 * i] = apow[i], table[2 * i + 1] = glh::mul(apow[i], 1ull << 32);
    }
    // one image of GateTab per unit: par | pih | apow | bias
    k->h_tab.assign(k->units.size(), {});
    for (size_t ui = 0; ui < k->units.size(); ui++) {
        const GateUnit &u = k->units[ui];
        std::vector<uint64_t> &img = k->h_tab[ui];
        img.reserve(10 + table.size() + (size_t)k->num_challenges * u.gates.size());
        img = {(uint64_t)(uintptr_t)wires, w_rs, w_es, (uint64_t)(uintptr_t)cs, c_rs, c_es, pih[0] % glh::P, pih[1] % glh::P, pih[2] % glh::P, pih[3] % glh::P};
        img.insert(img.end(), table.begin(), table.end());
        const size_t b0 = img.size();
        img.resize(b0 + (size_t)k->num_challenges * u.gates.size(), 0);
        if (u.biased)
            for (uint32_t c = 0; c < k->num_challenges; c++)
                for (size_t gi = 0; gi < u.gates.size(); gi++) {
                    uint64_t b = 0;
                    for (const auto &kb : u.bias[gi]) b = glh::add(b, glh::mul(apow[(size_t)c * k->num_constraints + kb.first], kb.second));
                    img[b0 + (size_t)c * u.gates.size() + gi] = b;
                }
        e = hipMemcpyAsync(u.d_tab, img.data(), img.size() * sizeof(uint64_t), hipMemcpyHostToDevice, stream);
        if (e != hipSuccess) return e;
    }
    const unsigned grid = (unsigned)((lde_size + 127) / 128);
    int accumulate = 0;
    for (const GateUnit &u : k->units) {
        void *args[] = {&wires, &w_rs, &w_es, &cs, &c_rs, &c_es, &lde_size, &out, &accumulate};
        e = hipModuleLaunchKernel(u.fn, grid, 1, 1, 128, 1, 1, 0, stream, args, nullptr);
        if (e != hipSuccess) return e;
        accumulate = 1;
    }
    e = hipEventRecord(k->done, stream);
    if (e != hipSuccess) return e;
    k->launched = true;
    return hipSuccess;
}

}  // namespace plonky2_hip
