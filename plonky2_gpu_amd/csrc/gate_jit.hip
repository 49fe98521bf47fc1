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
#include <sys/stat.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <sstream>
#include <vector>

#include "gl_field.h"

namespace plonky2_hip {

namespace {
const char *const GL_FIELD_SRC =
#include "build/gl_field_src.inc"
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

struct GateKernel {
    hipModule_t module = nullptr;
    hipFunction_t fn = nullptr;
    uint64_t *d_apow = nullptr;  // the module's g_apow[num_challenges][num_constraints]
    uint64_t *d_pih = nullptr;   // the module's g_pih[4]
    uint32_t num_challenges = 0, num_constraints = 0;
    uint32_t wires_needed = 0, constants_needed = 0;  // 1 + the largest wire / constant column any gate loads
    std::string source;
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
    for (uint32_t g = 0; g < num_gates; g++) {
        const uint32_t *d = gates + 6 * g;
        const uint32_t si = d[1], gs = d[2], ge = d[3], ps = d[4], pl = d[5];
        if ((uint64_t)ps + pl > num_instrs || si >= num_selectors || gs > ge) {
            *error = "gate descriptor out of range";
            return false;
        }
        uint32_t emitted = 0;
        for (uint32_t pc = ps; pc < ps + pl; pc++) {
            const uint16_t op = instrs[4 * pc], a = instrs[4 * pc + 2], b = instrs[4 * pc + 3];
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

static std::string generate_source(const uint16_t *instrs, uint32_t num_instrs, const uint32_t *gates, uint32_t num_gates,
                                   const uint64_t *imms, uint32_t num_imms, uint32_t num_selectors, uint32_t ngc, uint32_t nch,
                                   std::string *error) {
    std::ostringstream o;
    o << "#define GL_JIT 1\n" << GL_FIELD_SRC << "\n";
    o << "#define NCH " << nch << "\n#define NGC " << ngc << "\n";
    o << "struct GateSum { uint64_t v[NCH]; };\n";
    // alpha powers and the public-inputs hash live at link-time-constant addresses, so every read is a scalar
    // load (a pointer ARGUMENT of a non-inlined device function arrives in VGPRs and would be read per lane)
    o << "__constant__ uint64_t g_apow[NCH * NGC];\n__constant__ uint64_t g_pih[4];\n";
    for (uint32_t g = 0; g < num_gates; g++) {
        const uint32_t *d = gates + 6 * g;
        const uint32_t row = d[0], si = d[1], gs = d[2], ge = d[3], ps = d[4], pl = d[5];
        if (ps + pl > num_instrs || si >= num_selectors || gs > ge) {
            *error = "gate descriptor out of range";
            return "";
        }
        o << "static __device__ __noinline__ GateSum gate_" << g
          << "(const uint64_t* __restrict__ W, uint64_t wes, const uint64_t* __restrict__ C, uint64_t ces) {\n";
        // compute_filter (gates/gate.rs:261-268)
        o << "  const uint64_t s = C[" << si << " * ces];\n  uint64_t filt = 1;\n";
        for (uint32_t i = gs; i < ge; i++)
            if (i != row) o << "  filt = gl::mul(filt, gl::sub(" << i << "ull, s));\n";
        if (num_selectors > 1) o << "  filt = gl::mul(filt, gl::sub(0xFFFFFFFFull, s));\n";  // UNUSED_SELECTOR (selectors.rs:11)
        // the gate's constraints are reduced with powers of alpha lazily: one 192-bit column accumulator per
        // challenge, one reduction per gate (gl::DotAcc) instead of a multiply-reduce-add per constraint
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
                    o << "  acc" << q << "l += (uint64_t)(uint32_t)r" << ra << " * " << imms[b] << "u; acc" << q << "h += (uint64_t)(uint32_t)(r" << ra << " >> 32) * "
                      << imms[b] << "u;\n";
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
        o << "  GateSum out;\n  for (int c = 0; c < NCH; c++) out.v[c] = gl::mul(filt, gl::dot_finish(ga[c]));\n  return out;\n}\n";
    }
    o << "extern \"C\" __global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(4, 4))) void gate_constraints_kernel(const uint64_t* __restrict__ wires, uint64_t wrs, "
         "uint64_t wes, const uint64_t* __restrict__ cs, uint64_t crs, uint64_t ces, uint64_t lde_size, uint64_t* __restrict__ out) {\n"
         "  const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;\n  if (t >= lde_size) return;\n"
         "  const uint64_t* W = wires + t * wrs;\n  const uint64_t* C = cs + t * crs;\n"
         "  uint64_t acc[NCH];\n  for (int c = 0; c < NCH; c++) acc[c] = 0;\n";
    for (uint32_t g = 0; g < num_gates; g++)
        o << "  { GateSum s = gate_" << g << "(W, wes, C, ces); for (int c = 0; c < NCH; c++) acc[c] = gl::add(acc[c], s.v[c]); }\n";
    o << "  for (int c = 0; c < NCH; c++) out[(uint64_t)c * lde_size + t] = gl::canon(acc[c]);\n}\n";
    return o.str();
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
    k->source = generate_source(instrs, num_instrs, gates, num_gates, imms, num_imms, num_selectors, num_gate_constraints,
                                num_challenges, error);
    if (k->source.empty()) {
        delete k;
        return nullptr;
    }
    // On-disk cache (kernel_cache_dir()): the code object is keyed by a hash of the generated source, so a
    // circuit is compiled once per machine instead of once per process; the source is stored next to it for
    // inspection.
    // The key also covers what turns the same source into a different code object: the hiprtc version and the target.
    std::string cache_path;
    if (std::string dir = kernel_cache_dir(); !dir.empty()) {
        int rtc_major = 0, rtc_minor = 0;
        (void)hiprtcVersion(&rtc_major, &rtc_minor);
        const std::string salt = "hiprtc " + std::to_string(rtc_major) + "." + std::to_string(rtc_minor) + " " + JIT_ARCH + "\n";
        uint64_t h = 0xcbf29ce484222325ull;  // FNV-1a
        for (unsigned char ch : salt) h = (h ^ ch) * 0x100000001b3ull;
        for (unsigned char ch : k->source) h = (h ^ ch) * 0x100000001b3ull;
        char name[64];
        snprintf(name, sizeof name, "/gate_%016llx", (unsigned long long)h);
        cache_path = dir + name;
    }
    auto compile = [&](std::vector<char> &code) -> bool {
        hiprtcProgram prog;
        hiprtcResult r = hiprtcCreateProgram(&prog, k->source.c_str(), "gate_constraints.hip", 0, nullptr, nullptr);
        if (r != HIPRTC_SUCCESS) {
            *error = std::string("hiprtcCreateProgram: ") + hiprtcGetErrorString(r);
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
            *error = std::string("hiprtcCompileProgram: ") + hiprtcGetErrorString(r) + "\n" + log.substr(0, 4000);
            hiprtcDestroyProgram(&prog);
            return false;
        }
        size_t cs = 0;
        hiprtcGetCodeSize(prog, &cs);
        code.resize(cs);
        hiprtcGetCode(prog, code.data());
        hiprtcDestroyProgram(&prog);
        if (!cache_path.empty()) {
            // Several processes (one per GPU) may build the same circuit at once: each writes a file of its own and
            // renames it into place, so a reader sees either nothing or a whole code object.
            const std::string tmp = cache_path + ".tmp." + std::to_string((long long)getpid());
            std::ofstream(cache_path + ".hip." + std::to_string((long long)getpid())) << k->source;
            (void)rename((cache_path + ".hip." + std::to_string((long long)getpid())).c_str(), (cache_path + ".hip").c_str());
            bool written = false;
            {
                std::ofstream f(tmp, std::ios::binary);
                f.write(code.data(), (std::streamsize)code.size());
                f.flush();
                written = f.good();
            }
            if (!written || rename(tmp.c_str(), (cache_path + ".hsaco").c_str()) != 0) (void)remove(tmp.c_str());
        }
        return true;
    };
    auto load = [&](const std::vector<char> &code) -> hipError_t {
        hipError_t e = hipModuleLoadData(&k->module, code.data());
        if (e == hipSuccess) e = hipModuleGetFunction(&k->fn, k->module, "gate_constraints_kernel");
        size_t bytes = 0;
        if (e == hipSuccess) e = hipModuleGetGlobal(reinterpret_cast<hipDeviceptr_t *>(&k->d_apow), &bytes, k->module, "g_apow");
        if (e == hipSuccess) e = hipModuleGetGlobal(reinterpret_cast<hipDeviceptr_t *>(&k->d_pih), &bytes, k->module, "g_pih");
        return e;
    };
    std::vector<char> code;
    bool from_cache = false;
    if (!cache_path.empty()) {
        std::ifstream f(cache_path + ".hsaco", std::ios::binary);
        if (f) code.assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
        from_cache = !code.empty();
    }
    if (code.empty() && !compile(code)) {
        delete k;
        return nullptr;
    }
    hipError_t e = load(code);
    int ndev = 0;
    if (e != hipSuccess && from_cache && hipGetDeviceCount(&ndev) == hipSuccess && ndev > 0) {
        // a cached object that does not load (written by another ROCm, damaged): drop it and compile
        (void)hipGetLastError();
        if (k->module) (void)hipModuleUnload(k->module);
        k->module = nullptr;
        (void)remove((cache_path + ".hsaco").c_str());
        code.clear();
        if (!compile(code)) {
            delete k;
            return nullptr;
        }
        e = load(code);
    }
    if (e != hipSuccess) {
        *error = std::string("loading the compiled gate kernel: ") + hipGetErrorString(e);
        gate_kernel_destroy(k);
        return nullptr;
    }
    return k;
}

void gate_kernel_destroy(GateKernel *k) {
    if (!k) return;
    if (k->module) (void)hipModuleUnload(k->module);
    delete k;
}

uint32_t gate_kernel_num_challenges(const GateKernel *k) { return k->num_challenges; }
uint32_t gate_kernel_num_constraints(const GateKernel *k) { return k->num_constraints; }
uint32_t gate_kernel_wires_needed(const GateKernel *k) { return k->wires_needed; }
uint32_t gate_kernel_constants_needed(const GateKernel *k) { return k->constants_needed; }
const char *gate_kernel_source(const GateKernel *k) { return k->source.c_str(); }

hipError_t gate_kernel_launch(const GateKernel *k, const uint64_t *wires, uint64_t w_rs, uint64_t w_es, const uint64_t *cs,
                              uint64_t c_rs, uint64_t c_es, const uint64_t *alphas, const uint64_t pih[4], uint64_t lde_size,
                              uint64_t *out, hipStream_t stream) {
    std::vector<uint64_t> apow((size_t)k->num_challenges * k->num_constraints);
    for (uint32_t c = 0; c < k->num_challenges; c++) {
        uint64_t a = alphas[c] % glh::P, p = 1;
        for (uint32_t j = 0; j < k->num_constraints; j++) {
            apow[(size_t)c * k->num_constraints + j] = p;
            p = glh::mul(p, a);
        }
    }
    // pageable source: the copy has left the host buffer when hipMemcpyAsync returns
    hipError_t e = hipMemcpyAsync(k->d_apow, apow.data(), apow.size() * sizeof(uint64_t), hipMemcpyHostToDevice, stream);
    if (e != hipSuccess) return e;
    const uint64_t pi[4] = {pih[0] % glh::P, pih[1] % glh::P, pih[2] % glh::P, pih[3] % glh::P};
    e = hipMemcpyAsync(k->d_pih, pi, sizeof pi, hipMemcpyHostToDevice, stream);
    if (e != hipSuccess) return e;
    e = hipStreamSynchronize(stream);
    if (e != hipSuccess) return e;
    void *args[] = {&wires, &w_rs, &w_es, &cs, &c_rs, &c_es, &lde_size, &out};
    const unsigned grid = (unsigned)((lde_size + 127) / 128);
    return hipModuleLaunchKernel(k->fn, grid, 1, 1, 128, 1, 1, 0, stream, args, nullptr);
}

}  // namespace plonky2_hip
