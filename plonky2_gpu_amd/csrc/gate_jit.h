// gate_jit.h — circuit-specialised gate-constraint kernels compiled at run time (see gate_jit.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

namespace plonky2_hip {

struct GateKernel;  // opaque: hipModule + function + device table of alpha powers

// instrs: 4 x u16 per instruction (op, dst, a, b); gates: 6 x u32 per gate (row, selector_index, group_start,
// group_end, prog_start, prog_len) — the same encoding as GlGateInstr / GlGateDesc. Returns nullptr and fills
// `error` on failure (generated source that does not compile, hiprtc / module errors).
GateKernel *gate_kernel_build(const uint16_t *instrs, uint32_t num_instrs, const uint32_t *gates, uint32_t num_gates,
                              const uint64_t *imms, uint32_t num_imms, uint32_t num_selectors, uint32_t num_gate_constraints,
                              uint32_t num_challenges, std::string *error);
void gate_kernel_destroy(GateKernel *k);
uint32_t gate_kernel_num_challenges(const GateKernel *k);
uint32_t gate_kernel_num_constraints(const GateKernel *k);
uint32_t gate_kernel_wires_needed(const GateKernel *k);      // 1 + the largest wire index a gate loads
uint32_t gate_kernel_constants_needed(const GateKernel *k);  // 1 + the largest constants column a gate reads (selectors included)

// The checks shared by the compiled kernel and the interpreter's callers: descriptors in range, immediates in range, no gate
// emitting more than num_gate_constraints constraints; reports the wire / constant columns the programs need.
bool gate_programs_validate(const uint16_t *instrs, uint32_t num_instrs, const uint32_t *gates, uint32_t num_gates, uint32_t num_imms,
                            uint32_t num_selectors, uint32_t num_gate_constraints, uint32_t *wires_needed,
                            uint32_t *constants_needed, std::string *error);
const char *gate_kernel_source(const GateKernel *k);

// out[c*lde_size + t] = sum_k alpha_c^k * (sum_g filter_g * constraint_{g,k}) at the point held by leaf t, for
// t < lde_size. Element j of leaf t of the wires / constants_sigmas LDE is read at base[t*rs + j*es].
hipError_t gate_kernel_launch(const GateKernel *k, const uint64_t *wires, uint64_t w_rs, uint64_t w_es, const uint64_t *cs,
                              uint64_t c_rs, uint64_t c_es, const uint64_t *alphas, const uint64_t pih[4], uint64_t lde_size,
                              uint64_t *out, hipStream_t stream);

}  // namespace plonky2_hip
