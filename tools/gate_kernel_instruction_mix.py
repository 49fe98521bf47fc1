#!/usr/bin/env python3
"""Static instruction mix of the run-time compiled gate kernels (plonky2_gpu_amd/kernel_cache/*.hsaco, the code objects build()
precompiles for the ed25519 gate table) next to the operation mix of the register programs they were generated from and a
per-operation price list: how far the generated code is from what its programs cost on this ISA.
The table is compiled into a scratch cache first (18 s), so the figures are those of the current generator and knobs.
No GPU needed: llvm-objdump of the code objects.   python tools/gate_kernel_instruction_mix.py > profiles/r05_quotient_instruction_mix.json"""
import collections
import glob
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def compile_into(cache):
    """the ed25519 table through gl_gate_kernel_build with `cache` as the kernel cache: the generated sources and code objects of the
    CURRENT generator (and of its current knobs, e.g. PLONKY2_HIP_JIT_PEEPHOLE=0), whatever the in-tree cache holds"""
    import ctypes

    import numpy as np

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_ed25519_program as gen

    os.environ["PLONKY2_HIP_KERNEL_CACHE"] = cache
    if any(k in os.environ for k in ("PLONKY2_HIP_JIT_FUSE", "PLONKY2_HIP_JIT_PEEPHOLE", "PLONKY2_HIP_JIT_FUSE_GATES", "PLONKY2_HIP_JIT_PREFETCH", "PLONKY2_HIP_JIT_WAVES")):
        os.environ.setdefault("PLONKY2_HIP_LIBRARY", os.path.join(ROOT, "plonky2_gpu_amd", "libplonky2_hip_debug.so"))  # switches: diagnostic build only
    os.environ.setdefault("PLONKY2_HIP_JIT_FORK", "1")  # no HIP call has been made in this process
    from plonky2_gpu_amd import _lib

    instrs, descs, imms = gen.arrays()
    imms = np.array(imms, dtype=np.uint64)
    k = ctypes.c_void_p()
    try:
        _lib.call("gl_gate_kernel_build", instrs, instrs.size // 4, descs, descs.size // 6, imms, imms.size, 6, 231, 2, ctypes.byref(k))
        _lib.load().gl_gate_kernel_destroy(k)
    except _lib.Plonky2HipError as e:  # without a device the build stops at loading the module, after the cache write
        if "loading the compiled gate kernel" not in str(e):
            raise


# what the generated source calls, and what each costs in vector instructions (executed; the correction behind the never-taken
# branch of an operation is listed separately)
GENERATED = [("mul", r"gl::mul\(", 12, 3), ("mul_add_small", r"gl::mul_add_small<", 12, 3), ("add", r"gl::add\(", 4, 2), ("sub", r"gl::sub\(", 5, 3),
             ("add_small", r"gl::add_small<", 2, 3), ("sub_small", r"gl::sub_small<", 2, 3), ("mul_k", r"gl::mul_k<", 12, 3), ("add_k", r"gl::add_k<", 4, 2), ("emit (dot_term per challenge)", r"gl::dot_term2?\(", 16, 0),
             ("acc", r"gj_acc\(", 2, 0), ("accr (fold96)", r"gl::fold96\(", 7, 0), ("mulk", r"gl::mul_pow2<", 8, 0),
             ("load wire / constant", r"= [WCq]\[[0-9]", 2, 0), ("load immediate", r"= 0x[0-9a-f]+ull;", 2, 0), ("dot_finish", r"gl::dot_finish2?\(", 20, 0)]


def main():
    import tempfile

    from plonky2_gpu_amd import ed25519_circuit as ed
    from plonky2_gpu_amd import gate_program as gp

    names = ["LOAD_WIRE", "LOAD_CONST", "LOAD_PI", "LOAD_IMM", "ADD", "SUB", "MUL", "EMIT", "MULK", "ACC", "ACCR"]
    pool, ops = gp.ImmediatePool(), collections.Counter()
    for kind, param in ed.GATES:
        ops += collections.Counter(names[i[0]] for i in gp.build_gate(kind, param, pool))
    price = {"LOAD_WIRE": 2, "LOAD_CONST": 2, "LOAD_PI": 0, "LOAD_IMM": 2, "ADD": 4, "SUB": 5, "MUL": 12, "EMIT": 16, "MULK": 8, "ACC": 2, "ACCR": 7}
    cache = tempfile.mkdtemp(prefix="gate_mix_")
    compile_into(cache)
    generated = collections.Counter()
    for f in sorted(glob.glob(os.path.join(cache, "*.hip"))):
        body = open(f).read()
        starts = [body.find(m) for m in ("static __device__ __noinline__ GateSum gate_", 'extern "C" __global__')]
        body = body[min(i for i in starts if i >= 0):]
        for name, rx, _, _ in GENERATED:
            generated[name] += len(re.findall(rx, body))
    generated["dot_finish"] *= 2  # written once, in a loop over the two challenges
    mix, per_fn = collections.Counter(), {}
    for f in sorted(glob.glob(os.path.join(cache, "*.hsaco"))):
        cur = None
        for line in subprocess.run([OBJDUMP, "-d", f], capture_output=True, text=True).stdout.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
            if m:
                cur = m.group(1)
                per_fn.setdefault(cur, collections.Counter())
                continue
            m = re.match(r"^\s+([a-z]\w+)", line)
            if m and cur:
                mix[m.group(1)] += 1
                per_fn[cur][m.group(1)] += 1
    valu = sum(n for o, n in mix.items() if o.startswith("v_"))
    never = sum(generated[name] * rare for name, _, _, rare in GENERATED)
    floor = sum(price[k] * v for k, v in ops.items())
    generated_price = sum(generated[name] * cost for name, _, cost, _ in GENERATED)
    out = {"what": "ed25519 gate table (25 gates): register-program operations per LDE point, what the generator turns them into (its peephole pass "
                   "rewrites operations with small constants and the base-4 range checks), the vector instructions of the code objects, and a price list for both",
           "peephole": os.environ.get("PLONKY2_HIP_JIT_PEEPHOLE", "1") != "0",
           "program_operations": dict(ops), "program_operations_total": sum(ops.values()),
           "price_list_vector_instructions_per_program_operation": price, "programs_priced_as_written": floor,
           "generated_operations": dict(generated),
           "price_list_vector_instructions_per_generated_operation": {name: cost for name, _, cost, _ in GENERATED},
           "generated_operations_priced": generated_price,
           "static_vector_instructions": valu, "of_which_rare_path_corrections_never_executed": never,
           "executed_vector_instructions_estimate": valu - never,
           "executed_per_program_operation": (valu - never) / sum(ops.values()),
           "overhead_over_the_generated_operations_price": (valu - never) / generated_price - 1.0,
           "executed_over_programs_priced_as_written": (valu - never) / floor,
           "most_frequent_instructions": dict(mix.most_common(24)),
           "vector_instructions_per_gate_function": {k: sum(n for o, n in c.items() if o.startswith("v_")) for k, c in sorted(per_fn.items())}}
    print(json.dumps(out, indent=1))
    import shutil
    shutil.rmtree(cache, ignore_errors=True)


if __name__ == "__main__":
    main()
