/* prove_oracle.h — CPU oracle for everything ABOVE the commit (TEST INFRASTRUCTURE ONLY): the gate constraints,
 * the permutation argument, compute_quotient_polys, OpeningSet::new, PolynomialBatch::prove_openings / fri_proof and the
 * proof's wire format — a plain-C, OpenMP-threaded restatement of prove() (plonky2/src/plonk/prover.rs:40-233) from the
 * full witness on, so that whole proofs at the benchmark's shapes (2^18, 2^20 rows) are compared BYTE FOR BYTE and so
 * that bench.py's prove() leg has a CPU baseline that is a real prove(). Built on gl_oracle.c (field, NTT, Poseidon,
 * Merkle). Nothing in the product may include, link or call it.
 *
 * Parity status: the reference holds no proof fixtures and cannot be built here, so this file is pinned
 * (a) against the independent Python restatement (oracle/prove_ref.py etc.: identical proof bytes on the small circuits,
 * tests/test_oracle_prove_c.py), whose verifier re-derives every challenge and checks vanishing(zeta) = Z_H(zeta) t(zeta),
 * and (b) at the quotient stage against the reference's own device kernel compiled for gfx950 (oracle/_ref,
 * tests/test_gpu_reference_kernels.py). FRI and the wire format remain the builder's reading of the Rust: "parity
 * unpinned" for those two rows, as DESIGN.md §6 says.
 *
 * One deliberate deviation from the reference, the same as everywhere else in this repository: the proof-of-work witness
 * is the SMALLEST one (the reference's rayon find_any returns an arbitrary one, fri/prover.rs:151-163).
 */
#ifndef PROVE_ORACLE_H
#define PROVE_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* gate kinds and their parameters (params[0..2]):
 *   NOOP, PUBLIC_INPUT, POSEIDON, POSEIDON_MDS: none    CONSTANT {num_consts}     ARITHMETIC {num_ops}
 *   BASE_SUM {B, num_limbs}       U32_ADD_MANY {num_addends, num_ops}             U32_ARITHMETIC / U32_SUBTRACTION {num_ops}
 *   U32_RANGE_CHECK {num_input_limbs}   COMPARISON {num_bits, num_chunks}         RANDOM_ACCESS {bits, num_copies, num_extra_constants}
 *   ARITHMETIC_EXTENSION / MUL_EXTENSION {num_ops}      REDUCING / REDUCING_EXTENSION {num_coeffs}
 *   EXPONENTIATION {num_power_bits}     LOW_DEGREE_INTERPOLATION / HIGH_DEGREE_INTERPOLATION {subgroup_bits} */
enum glo_gate_kind {
    GLO_GATE_NOOP = 0, GLO_GATE_CONSTANT, GLO_GATE_PUBLIC_INPUT, GLO_GATE_ARITHMETIC, GLO_GATE_BASE_SUM, GLO_GATE_U32_ADD_MANY,
    GLO_GATE_U32_ARITHMETIC, GLO_GATE_U32_SUBTRACTION, GLO_GATE_U32_RANGE_CHECK, GLO_GATE_COMPARISON, GLO_GATE_RANDOM_ACCESS,
    GLO_GATE_POSEIDON,
    GLO_GATE_ARITHMETIC_EXTENSION, GLO_GATE_MUL_EXTENSION, GLO_GATE_REDUCING, GLO_GATE_REDUCING_EXTENSION, GLO_GATE_EXPONENTIATION,
    GLO_GATE_POSEIDON_MDS, GLO_GATE_LOW_DEGREE_INTERPOLATION, GLO_GATE_HIGH_DEGREE_INTERPOLATION,
    GLO_GATE_KINDS
};
typedef struct glo_gate {
    uint32_t kind, params[3], selector_index;
} glo_gate;

/* CommonCircuitData + the prover-side part of ProverOnlyCircuitData (plonk/circuit_data.rs) */
typedef struct glo_circuit_desc {
    uint32_t degree_bits, num_wires, num_routed_wires, num_constants, num_challenges, quotient_degree_factor;
    uint32_t num_gate_constraints;
    uint32_t rate_bits, cap_height, proof_of_work_bits, num_query_rounds, num_reductions;
    const uint32_t *reduction_arity_bits; /* num_reductions */
    uint32_t hiding;                      /* FriParams::hiding */
    const uint64_t *k_is;                 /* num_routed_wires */
    const uint64_t *constants;            /* [num_constants][n] value columns, selectors first */
    const uint64_t *sigmas;               /* [num_routed_wires][n] value columns */
    const glo_gate *gates;
    uint32_t num_gates, num_selectors;
    const uint32_t *group_bounds;         /* [2*num_selectors]: selectors_info.groups[s] = [b[2s], b[2s+1]) */
    const uint64_t *circuit_digest;       /* 4, or NULL: derived as circuit_builder.rs:915-927 (empty domain separator) */
} glo_circuit_desc;

/* Unfiltered constraints of one gate at one point over the base field (Gate::eval_unfiltered_base_one of the gate's file).
 * consts = local_constants after the selector prefix, w = local_wires, pih = public_inputs_hash. Returns the number of
 * constraints written to out, or -1 for an unknown kind. */
int glo_gate_constraints(const glo_gate *g, const uint64_t *consts, const uint64_t *w, const uint64_t pih[4], uint64_t *out);
int glo_gate_num_constraints(const glo_gate *g);

/* evaluate_gate_constraints_base_batch for one point (plonk/vanishing_poly.rs:267-306, gates/gate.rs:109-150, 261-268):
 * out[num_gate_constraints] = sum over gates of filter * constraint. */
void glo_evaluate_gate_constraints(const glo_circuit_desc *c, const uint64_t *local_constants, const uint64_t *local_wires,
                                   const uint64_t pih[4], uint64_t *out);

/* the circuit with its preprocessed commitment (constants || sigmas) */
void *glo_circuit_new(const glo_circuit_desc *desc, int n_threads);
void glo_circuit_free(void *circuit);
void glo_circuit_info(const void *circuit, uint64_t digest[4], uint64_t *constants_sigmas_cap /* 4 << cap_height */);

/* optional taps on the intermediate objects (any pointer may be NULL) */
typedef struct glo_prove_trace {
    uint64_t *betas, *gammas, *alphas; /* num_challenges each */
    uint64_t *zeta;                    /* 2 */
    uint64_t *zs_partial_products;     /* [num_challenges*(1+num_prods)][n] values, every Z first */
    uint64_t *quotient_polys;          /* [num_challenges][n << log2_ceil(qdf)] coefficients */
    uint64_t *wires_cap, *zs_cap, *quotient_cap; /* 4 << cap_height each */
    double stage_seconds[8];           /* wires commit, partial products, zs commit, quotient, quotient commit, openings, FRI, total */
} glo_prove_trace;

/* prove() (plonk/prover.rs:40-233) from the full witness on. wires [num_wires][n] value columns; salts (hiding circuits
 * only, else NULL) [3][4][n_ext] in LEAF order. The proof comes back in the reference's wire format
 * (write_proof_with_public_inputs, util/serialization.rs:674-689), malloc'ed: free with glo_bytes_free.
 * Returns 0, or a negative code with a message in err: -1 bad arguments, -2 out of memory,
 * -3 "Quotient has failed, the vanishing polynomial is not divisible by Z_H", -4 "Opening point is in the subgroup." */
int glo_prove(const void *circuit, const uint64_t *wires, const uint64_t *public_inputs, uint32_t num_public_inputs,
              const uint64_t *salts, uint8_t **proof, size_t *proof_len, glo_prove_trace *trace, int n_threads, char *err,
              size_t err_len);
void glo_bytes_free(uint8_t *p);

#ifdef __cplusplus
}
#endif
#endif
