// prove_example.cpp — a compiled host that proves with THREE library calls and no Python: what a Rust
// `cuda/`-replacement crate would do through FFI. It reads a circuit + witness
// file (layout below, written by tests/test_cpp_prove.py) whose gates are given as plonky2's gate list (kinds and
// parameters), has the library emit their register programs (gl_gate_programs_emit), calls gl_circuit_create and gl_prove, and
// writes the proof in the reference's wire format (plonky2/src/util/serialization.rs:674-689).
//
// file = little-endian u64 stream:
//   header[20]: magic 0x706c6f6e6b7932, degree_bits, num_wires, num_routed_wires, num_constants, num_challenges,
//               quotient_degree_factor, num_gate_constraints, rate_bits, cap_height, proof_of_work_bits,
//               num_query_rounds, num_reductions, num_selectors, num_gates, num_instrs, num_immediates,
//               num_public_inputs, compile_gates, gate_list_mode
//   reduction_arity_bits[num_reductions], k_is[num_routed], constants[num_constants * n], sigmas[num_routed * n],
//   gate_list_mode = 0: instrs[num_instrs] (one u64 = {op, dst, a, b} as four u16), gates[num_gates * 3] (six u32),
//                       immediates[num_immediates]   — register programs made elsewhere
//   gate_list_mode = 1: gate_specs[num_gates * 5] (kind, three parameters, selector index), group_bounds[num_selectors * 2]
//                       — the circuit's GATE LIST; the register programs are emitted here, by gl_gate_programs_emit
//   wires[num_wires * n], public_inputs[num_public_inputs]
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "plonky2_hip.h"

static void check(GlError e, const char *what) {
    if (e.code != 0) {
        fprintf(stderr, "%s failed: %d %s\n", what, e.code, e.message ? e.message : cudaGetErrorString(e.code));
        if (e.message) free(e.message);
        exit(2);
    }
}

int main(int argc, char **argv) {
    if (argc != 3) {
        fprintf(stderr, "usage: %s circuit_and_witness.bin proof.bin\n", argv[0]);
        return 1;
    }
    FILE *f = fopen(argv[1], "rb");
    if (!f) return perror(argv[1]), 1;
    fseek(f, 0, SEEK_END);
    const long bytes = ftell(f);
    fseek(f, 0, SEEK_SET);
    std::vector<uint64_t> d((size_t)bytes / 8);
    if (fread(d.data(), 8, d.size(), f) != d.size()) return fprintf(stderr, "short read\n"), 1;
    fclose(f);
    if (d.size() < 20 || d[0] != 0x706c6f6e6b7932ull) return fprintf(stderr, "bad magic\n"), 1;
    const uint64_t *h = d.data();
    const uint64_t n = 1ull << h[1];
    size_t pos = 20;
    auto take = [&](uint64_t count) {
        const uint64_t *p = d.data() + pos;
        pos += count;
        if (pos > d.size()) {
            fprintf(stderr, "file too short\n");
            exit(1);
        }
        return p;
    };
    std::vector<uint32_t> arity;
    for (const uint64_t *p = take(h[12]), *e = p + h[12]; p < e; p++) arity.push_back((uint32_t)*p);
    GlCircuitDesc desc = {};
    desc.struct_size = sizeof desc;
    desc.degree_bits = (uint32_t)h[1], desc.num_wires = (uint32_t)h[2], desc.num_routed_wires = (uint32_t)h[3];
    desc.num_constants = (uint32_t)h[4], desc.num_challenges = (uint32_t)h[5], desc.quotient_degree_factor = (uint32_t)h[6];
    desc.num_gate_constraints = (uint32_t)h[7];
    desc.fri.rate_bits = (uint32_t)h[8], desc.fri.cap_height = (uint32_t)h[9], desc.fri.proof_of_work_bits = (uint32_t)h[10];
    desc.fri.num_query_rounds = (uint32_t)h[11], desc.fri.num_reductions = (uint32_t)h[12];
    desc.fri.reduction_arity_bits = arity.data();
    desc.num_selectors = (uint32_t)h[13], desc.num_gates = (uint32_t)h[14], desc.num_instrs = (uint32_t)h[15];
    desc.num_immediates = (uint32_t)h[16];
    const uint32_t num_public_inputs = (uint32_t)h[17];
    desc.compile_gates = (int)h[18];
    desc.h_k_is = take(h[3]);
    desc.h_constants = take(h[4] * n);
    desc.h_sigmas = take(h[3] * n);
    GlGatePrograms programs = {};
    if (h[19] == 1) {
        // the gate list of the circuit: kinds and parameters as plonky2 names them; the programs come from the library
        std::vector<GlGateSpec> specs(h[14]);
        const uint64_t *sp = take(h[14] * 5);
        for (uint64_t g = 0; g < h[14]; g++) {
            specs[g].kind = (uint32_t)sp[5 * g];
            for (int k = 0; k < 3; k++) specs[g].params[k] = (uint32_t)sp[5 * g + 1 + k];
            specs[g].selector_index = (uint32_t)sp[5 * g + 4];
        }
        std::vector<uint32_t> bounds;
        for (const uint64_t *p = take(h[13] * 2), *e = p + h[13] * 2; p < e; p++) bounds.push_back((uint32_t)*p);
        check(gl_gate_programs_emit(specs.data(), (uint32_t)specs.size(), bounds.data(), (uint32_t)h[13], &programs), "gl_gate_programs_emit");
        if (programs.num_gate_constraints != desc.num_gate_constraints) return fprintf(stderr, "num_gate_constraints: file says %u, the gate list gives %u\n", desc.num_gate_constraints, programs.num_gate_constraints), 1;
        desc.h_instrs = programs.instrs, desc.num_instrs = programs.num_instrs;
        desc.h_gates = programs.gates, desc.num_gates = programs.num_gates;
        desc.h_immediates = programs.immediates, desc.num_immediates = programs.num_immediates;
    } else {
        desc.h_instrs = reinterpret_cast<const GlGateInstr *>(take(h[15]));
        desc.h_gates = reinterpret_cast<const GlGateDesc *>(take(h[14] * 3));
        desc.h_immediates = take(h[16]);
    }
    const uint64_t *wires = take(h[2] * n);
    const uint64_t *public_inputs = take(num_public_inputs);
    desc.h_circuit_digest = nullptr;  // derived by the library (circuit_builder.rs:915-927)

    if (gl_device_count() <= 0) return fprintf(stderr, "no HIP device\n"), 3;
    void *ctx = gl_ctx_create(0);
    if (!ctx) return fprintf(stderr, "gl_ctx_create failed\n"), 3;
    void *circuit = nullptr;
    check(gl_circuit_create(&desc, &circuit, ctx), "gl_circuit_create");
    void *d_wires = nullptr;
    check(gl_malloc(&d_wires, h[2] * n * 8), "gl_malloc");
    check(gl_memcpy_h2d(d_wires, wires, h[2] * n * 8, ctx), "gl_memcpy_h2d");
    uint8_t *proof = nullptr;
    uint64_t proof_len = 0;
    check(gl_prove(circuit, static_cast<const uint64_t *>(d_wires), public_inputs, num_public_inputs, &proof, &proof_len, nullptr, ctx),
          "gl_prove");
    uint64_t digest[4];
    check(gl_circuit_info(circuit, digest, nullptr), "gl_circuit_info");
    FILE *o = fopen(argv[2], "wb");
    if (!o || fwrite(proof, 1, proof_len, o) != proof_len) return perror(argv[2]), 1;
    fclose(o);
    printf("PROOF_BYTES %llu\nDIGEST %llu %llu %llu %llu\n", (unsigned long long)proof_len, (unsigned long long)digest[0],
           (unsigned long long)digest[1], (unsigned long long)digest[2], (unsigned long long)digest[3]);
    gl_bytes_free(proof);
    gl_gate_programs_free(&programs);
    check(gl_free(d_wires), "gl_free");
    gl_circuit_destroy(circuit);
    gl_ctx_destroy(ctx);
    return 0;
}
