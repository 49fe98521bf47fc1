"""The one circuit the reference's `compute_quotient_polys` symbol is compiled for: the plonky2-ed25519
signature circuit. These are the FACTS its CUDA kernel hard-wires (cuda/plonky2_gpu.cu:666-689 for the
shape and the public-inputs hash, cuda/plonky2_gpu_impl.cuh:597-685 for the gate table and selector groups);
the generic path (gl_compute_quotient_polys / gl_prove) takes all of this as arguments instead.

tools/gen_ed25519_program.py turns the gate table into csrc/ed25519_gate_program.inc, from which the
library's own `compute_quotient_polys` builds its kernel."""

# (kind, parameters) in gate-table order == the `row` the selector filters are computed from
GATES = [
    ("noop", None),                # 0
    ("constant", 2),               # 1  num_consts
    ("public_input", None),        # 2
    ("base_sum", (2, 32)),         # 3  (base, num_limbs)
    ("base_sum", (2, 63)),         # 4
    ("arithmetic", 20),            # 5  num_ops
    ("base_sum", (4, 16)),         # 6
    ("comparison", (32, 16)),      # 7  (num_bits, num_chunks)
    ("u32_add_many", (0, 11)),     # 8  (num_addends, num_ops)
    ("u32_add_many", (11, 5)),     # 9
    ("u32_add_many", (13, 5)),     # 10
    ("u32_add_many", (15, 4)),     # 11
    ("u32_add_many", (16, 4)),     # 12
    ("u32_add_many", (2, 10)),     # 13
    ("u32_add_many", (3, 9)),      # 14
    ("u32_add_many", (5, 9)),      # 15
    ("u32_add_many", (7, 8)),      # 16
    ("u32_add_many", (9, 6)),      # 17
    ("u32_arithmetic", 6),         # 18 num_ops
    ("u32_range_check", 0),        # 19 num_input_limbs
    ("u32_range_check", 1),        # 20
    ("u32_range_check", 8),        # 21
    ("u32_subtraction", 11),       # 22 num_ops
    ("random_access", (4, 4, 2)),  # 23 (bits, num_copies, num_extra_constants)
    ("poseidon", None),            # 24
]
GROUPS = [(0, 6), (6, 11), (11, 16), (16, 21), (21, 24), (24, 25)]  # selectors_info.groups
SELECTOR_INDICES = [0] * 6 + [1] * 5 + [2] * 5 + [3] * 5 + [4] * 3 + [5]

NUM_CHALLENGES = 2
NUM_GATE_CONSTRAINTS = 231
NUM_CONSTANTS = 8
NUM_ROUTED_WIRES = 80
NUM_WIRES = 234                       # wires_commitment_leaf_len
QUOTIENT_DEGREE_FACTOR = 8
NUM_PARTIAL_PRODUCTS = 9
CONSTANTS_SIGMAS_LEAF_LEN = 88        # 8 constants + 80 sigmas
ZS_PARTIAL_PRODUCTS_LEAF_LEN = 20     # 2 x (1 Z + 9 partial products)
RATE_BITS = 3
COSET_SHIFT = 7

# The reference passes the hash of ONE proof's public inputs as a compiled-in constant
# (cuda/plonky2_gpu.cu:686-689); the symbol here starts from the same value and
# gl_reference_set_public_inputs_hash() replaces it for any other instance of the circuit.
REFERENCE_PUBLIC_INPUTS_HASH = [0x672C5E6C12AD3476, 0xCA5C2E49ACFAD27E, 0x296BE18388D15F70, 0x66B42E146A70D96D]
