"""tests/cpp/prove_example.cpp — a compiled host that proves through gl_circuit_create + gl_prove only.
CPU: it compiles, links and refuses to run without a device. GPU: its proof bytes equal the oracle's."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "cpp", "prove_example")
MAGIC = 0x706C6F6E6B7932


def build_binary():
    lib_dir = os.path.join(ROOT, "plonky2_gpu_amd")
    if not os.path.exists(os.path.join(lib_dir, "libplonky2_hip.so")):
        import __graft_entry__ as g

        g.build()
    cmd = ["g++", "-std=c++17", "-O2", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "prove_example.cpp"),
           "-L", lib_dir, "-lplonky2_hip", "-Wl,-rpath," + lib_dir, "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib",
           "-o", BIN]
    subprocess.check_call(cmd)


def write_circuit_file(path, circuit, wires, public_inputs, compile_gates):
    """the file layout documented at the top of prove_example.cpp"""
    from plonky2_gpu_amd import gate_program as gp

    pool = gp.ImmediatePool()
    programs = [gp.build_gate(kind, param, pool) for kind, param in circuit["gates"]]
    instrs, descs = gp.pack_program(programs, circuit["selector_indices"], circuit["groups"])
    instrs = np.ascontiguousarray(instrs, dtype=np.uint16).reshape(-1, 4)
    descs = np.ascontiguousarray(descs, dtype=np.uint32).reshape(-1, 6)
    fp = circuit["fri_params"]
    header = [MAGIC, circuit["degree_bits"], circuit["num_wires"], circuit["num_routed_wires"], circuit["num_constants"],
              circuit["num_challenges"], circuit["quotient_degree_factor"], circuit["num_gate_constraints"], fp["rate_bits"],
              fp["cap_height"], fp["proof_of_work_bits"], fp["num_query_rounds"], len(fp["reduction_arity_bits"]), len(circuit["groups"]),
              len(programs), instrs.shape[0], len(pool.values), len(public_inputs), 1 if compile_gates else 0, 0]
    u64 = lambda a: np.ascontiguousarray(a, dtype=np.uint64).reshape(-1)  # noqa: E731
    parts = [u64(header), u64(fp["reduction_arity_bits"]), u64(circuit["k_is"]), u64(circuit["constants"]), u64(circuit["sigmas"]),
             instrs.reshape(-1).view(np.uint64), descs.reshape(-1).view(np.uint64), u64(pool.values), u64(wires), u64(public_inputs)]
    with open(path, "wb") as f:
        for p in parts:
            f.write(p.tobytes())


def test_cpp_prove_example_compiles_links_and_needs_a_device(tmp_path):
    import plonky2_gpu_amd as pg

    build_binary()
    assert os.path.exists(BIN)
    if pg.load().gl_device_count() > 0:
        pytest.skip("a GPU is visible")
    from plonk_instance import make_circuit

    circuit, wires, pis = make_circuit(4, seed=31)
    src = tmp_path / "c.bin"
    write_circuit_file(src, circuit, wires, pis, True)
    p = subprocess.run([BIN, str(src), str(tmp_path / "proof.bin")], capture_output=True, text=True, timeout=120)
    assert p.returncode == 3 and "no HIP device" in p.stderr  # loud, no CPU fallback


@pytest.mark.gpu
@pytest.mark.parametrize("which,compile_gates", [("mini", True), ("full", False)])
def test_cpp_host_proof_bytes_equal_the_oracle(tmp_path, which, compile_gates):
    from oracle import prove_ref, serialize_ref
    from plonk_instance import make_circuit, make_full_circuit

    build_binary()
    circuit, wires, pis = make_full_circuit(4, seed=4) if which == "full" else make_circuit(5, seed=31, two_groups=True, arity_bits=(3,))
    src, out = tmp_path / "c.bin", tmp_path / "proof.bin"
    write_circuit_file(src, circuit, wires, pis, compile_gates)
    p = subprocess.run([BIN, str(src), str(out)], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr
    lines = {l.split()[0]: [int(x) for x in l.split()[1:]] for l in p.stdout.splitlines() if l}
    assert lines["DIGEST"] == circuit["circuit_digest"]
    data = out.read_bytes()
    assert lines["PROOF_BYTES"] == [len(data)]
    assert data == serialize_ref.proof_bytes(prove_ref.prove(circuit, wires, pis))
