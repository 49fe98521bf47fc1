"""plonky2_gpu_amd — MI355X-native prover hot path for plonky2 (NTT/LDE, Poseidon Merkle caps,
PolynomialBatch commit) behind the reference's extern "C" boundary.

The product is the HIP library `libplonky2_hip.so` (C ABI: include/plonky2_hip.h). This package is
the thin host-side mirror of the reference's operator interface for this path
(`fft_with_options` / `ifft_with_options`, `MerkleTree::new/prove`,
`PolynomialBatch::from_values/from_coeffs/get_lde_values`) used by the tests and the bench.
No bulk field arithmetic happens on the CPU (only scalar transcript glue such as powers of a
challenge); without the HIP library it raises.
"""
from ._lib import GL_E_INVALID, GL_E_UNSUPPORTED, Plonky2HipError, load  # noqa: F401
from .device import Context, DeviceBuffer, Event, PinnedArray  # noqa: F401
from .fft import coset_fft, coset_ifft, coset_lde_bit_reversed, fft_with_options, ifft_with_options  # noqa: F401
from .merkle_tree import MerkleTree  # noqa: F401
from .polynomial_batch import PolynomialBatch  # noqa: F401
from .challenger import Challenger  # noqa: F401
from . import serialization  # noqa: F401
from .fri import prove_openings  # noqa: F401
from .prover import CircuitData, GateProgram, NativeCircuit, all_wires_permutation_partial_products, compute_quotient_polys, prove  # noqa: F401
from .prover import reference_compute_quotient_polys, reference_set_public_inputs_hash  # noqa: F401

P = 0xFFFFFFFF00000001
COSET_SHIFT = 7  # F::coset_shift(), field/src/types.rs:431-433
SALT_SIZE = 4  # plonky2/src/fri/oracle.rs:41
