#!/usr/bin/env python3
"""Static instruction mix of the run-time compiled gate kernels (plonky2_gpu_amd/kernel_cache/*.hsaco, the code objects build()
precompiles for the ed25519 gate table) next to the operation mix of the register programs they were generated from and a
per-operation price list: how far the generated code is from what its programs cost on this ISA.
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


def main():
    from plonky2_gpu_amd import ed25519_circuit as ed
    from plonky2_gpu_amd import gate_program as gp

    names = ["LOAD_WIRE", "LOAD_CONST", "LOAD_PI", "LOAD_IMM", "ADD", "SUB", "MUL", "EMIT", "MULK", "ACC", "ACCR"]
    pool, ops = gp.ImmediatePool(), collections.Counter()
    for kind, param in ed.GATES:
        ops += collections.Counter(names[i[0]] for i in gp.build_gate(kind, param, pool))
    price = {"LOAD_WIRE": 2, "LOAD_CONST": 2, "LOAD_PI": 0, "LOAD_IMM": 2, "ADD": 4, "SUB": 5, "MUL": 12, "EMIT": 16, "MULK": 8, "ACC": 2, "ACCR": 7}
    rare = {"ADD": 2, "SUB": 3, "MUL": 3}  # instructions of the correction behind the never-taken branch of each operation
    mix, per_fn = collections.Counter(), {}
    for f in sorted(glob.glob(os.path.join(ROOT, "plonky2_gpu_amd", "kernel_cache", "*.hsaco"))):
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
    never = sum(rare[k] * ops[k] for k in rare)
    floor = sum(price[k] * v for k, v in ops.items())
    out = {"what": "ed25519 gate table (25 gates): register-program operations per LDE point, the vector instructions of the generated code objects, and a per-operation price list",
           "program_operations": dict(ops), "program_operations_total": sum(ops.values()),
           "static_vector_instructions": valu, "of_which_rare_path_corrections_never_executed": never,
           "executed_vector_instructions_estimate": valu - never,
           "price_list_vector_instructions_per_operation": price, "priced_total": floor,
           "executed_per_program_operation": (valu - never) / sum(ops.values()), "priced_per_program_operation": floor / sum(ops.values()),
           "overhead_over_price_list": (valu - never) / floor - 1.0,
           "most_frequent_instructions": dict(mix.most_common(24)),
           "vector_instructions_per_gate_function": {k: sum(n for o, n in c.items() if o.startswith("v_")) for k, c in sorted(per_fn.items())}}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
