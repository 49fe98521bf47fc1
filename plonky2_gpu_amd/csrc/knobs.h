// knobs.h — the A/B switches of the kernels exist only in the DIAGNOSTIC build of the library.
//
// `make debug` compiles the sources that read them with -DPLONKY2_DEBUG_KNOBS into plonky2_gpu_amd/libplonky2_hip_debug.so (the
// Python host loads it with PLONKY2_HIP_LIBRARY=<path>); there debug_knob() is getenv(). In the product library it returns
// nullptr for every name: no environment variable changes which kernel runs, and the strings are not even in the binary
// (tests/test_abi.py). The variables: PLONKY2_NTT_DIRECT / _KERNEL / _WIDE / _XCD / _WG_PER_CU / _CHUNK_COLS, PLONKY2_TRANSPOSE,
// PLONKY2_COMMIT_PIPELINE, PLONKY2_DROP_STREAM2_WAIT, and the gate-kernel generator's PLONKY2_HIP_JIT_FUSE / _PEEPHOLE / _FUSE_GATES /
// _PREFETCH / _WAVES / _UNITS. Operational settings are not knobs and stay in the product:
// PLONKY2_HIP_KERNEL_CACHE (where compiled gate kernels are kept), PLONKY2_HIP_JIT_FORK (compile units in forked children),
// PLONKY2_HIP_REFERENCE_IN_PLACE (no staging buffer).
#pragma once
#include <stdlib.h>

namespace plonky2_hip {
#ifdef PLONKY2_DEBUG_KNOBS
#define PLONKY2_KNOB(name) getenv(name)
#else
#define PLONKY2_KNOB(name) (static_cast<const char *>(nullptr))
#endif
}  // namespace plonky2_hip
