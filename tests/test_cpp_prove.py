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


def write_circuit_file(path, circuit, wires, public_inputs, compile_gates, gate_list=True):
    """the file layout documented at the top of prove_example.cpp. gate_list=True: the file carries the circuit's gate LIST
    (kind, parameters, selector index) and the C++ host has the library emit the register programs (gl_gate_programs_emit) — no
    Python-made program travels; gate_list=False: programs packed here by plonky2_gpu_amd/gate_program.py."""
    from plonky2_gpu_amd import _lib, gate_program as gp

    fp = circuit["fri_params"]
    u64 = lambda a: np.ascontiguousarray(a, dtype=np.uint64).reshape(-1)  # noqa: E731
    if gate_list:
        specs = []
        for (kind, param), si in zip(circuit["gates"], circuit["selector_indices"]):
            ps = [] if param is None else ([param] if isinstance(param, int) else list(param))
            specs += [_lib.GATE_KINDS[kind]] + (ps + [0, 0, 0])[:3] + [si]
        gate_parts = [u64(specs), u64([b for g in circuit["groups"] for b in g])]
        n_instrs = n_imms = 0
    else:
        pool = gp.ImmediatePool()
        programs = [gp.build_gate(kind, param, pool) for kind, param in circuit["gates"]]
        instrs, descs = gp.pack_program(programs, circuit["selector_indices"], circuit["groups"])
        instrs = np.ascontiguousarray(instrs, dtype=np.uint16).reshape(-1, 4)
        descs = np.ascontiguousarray(descs, dtype=np.uint32).reshape(-1, 6)
        gate_parts = [instrs.reshape(-1).view(np.uint64), descs.reshape(-1).view(np.uint64), u64(pool.values)]
        n_instrs, n_imms = instrs.shape[0], len(pool.values)
    header = [MAGIC, circuit["degree_bits"], circuit["num_wires"], circuit["num_routed_wires"], circuit["num_constants"],
              circuit["num_challenges"], circuit["quotient_degree_factor"], circuit["num_gate_constraints"], fp["rate_bits"],
              fp["cap_height"], fp["proof_of_work_bits"], fp["num_query_rounds"], len(fp["reduction_arity_bits"]), len(circuit["groups"]),
              len(circuit["gates"]), n_instrs, n_imms, len(public_inputs), 1 if compile_gates else 0, 1 if gate_list else 0]
    parts = [u64(header), u64(fp["reduction_arity_bits"]), u64(circuit["k_is"]), u64(circuit["constants"]), u64(circuit["sigmas"])] + gate_parts + \
        [u64(wires), u64(public_inputs)]
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
@pytest.mark.parametrize("which,compile_gates,gate_list", [("mini", True, True), ("full", False, True), ("full", True, True), ("mini", True, False)])
def test_cpp_host_proof_bytes_equal_the_oracle(tmp_path, which, compile_gates, gate_list):
    """`full` is the 13-gate circuit with every gate KIND of the ed25519 list at other parameters than the compiled-in table's:
    with gate_list the C++ host describes it by kinds and parameters only, and the library emits the programs."""
    from oracle import prove_ref, serialize_ref
    from plonk_instance import make_circuit, make_full_circuit

    build_binary()
    circuit, wires, pis = make_full_circuit(4, seed=4) if which == "full" else make_circuit(5, seed=31, two_groups=True, arity_bits=(3,))
    src, out = tmp_path / "c.bin", tmp_path / "proof.bin"
    write_circuit_file(src, circuit, wires, pis, compile_gates, gate_list)
    p = subprocess.run([BIN, str(src), str(out)], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr
    lines = {l.split()[0]: [int(x) for x in l.split()[1:]] for l in p.stdout.splitlines() if l}
    assert lines["DIGEST"] == circuit["circuit_digest"]
    data = out.read_bytes()
    assert lines["PROOF_BYTES"] == [len(data)]
    assert data == serialize_ref.proof_bytes(prove_ref.prove(circuit, wires, pis))
