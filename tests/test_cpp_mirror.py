"""include/plonky2_hip.hpp — the compiled-host mirror of PolynomialBatch / MerkleTree over the C ABI.
CPU: it compiles and links against the library. GPU: it reproduces the golden commit fixture."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "cpp", "host_mirror_test")


def build_binary():
    lib_dir = os.path.join(ROOT, "plonky2_gpu_amd")
    if not os.path.exists(os.path.join(lib_dir, "libplonky2_hip.so")):
        import __graft_entry__ as g

        g.build()
    cmd = ["g++", "-std=c++17", "-O2", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "host_mirror_test.cpp"),
           "-L", lib_dir, "-lplonky2_hip", "-Wl,-rpath," + lib_dir, "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib",
           "-o", BIN]
    subprocess.check_call(cmd)


def test_cpp_mirror_compiles_and_links():
    build_binary()
    assert os.path.exists(BIN)


@pytest.mark.gpu
def test_cpp_mirror_reproduces_golden_commit(tmp_path, oracle):
    build_binary()
    g = np.load(os.path.join(ROOT, "tests", "golden", "commit_p135_2e6_r3_h4.npz"))
    vals = tmp_path / "values.bin"
    g["values"].astype(np.uint64).tofile(vals)
    p = subprocess.run([BIN, str(vals), "135", "3", "4", str(tmp_path / "out")], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    lines = {l.split()[0]: [int(x) for x in l.split()[1:]] for l in p.stdout.splitlines() if l and not l.startswith("CAP_TOO_BIG")}
    assert lines["CAP"] == g["cap"].reshape(-1).tolist()
    # get_lde_values(3) = leaf bitrev(3)
    bits = 6 + 3
    rev = int(f"{3:0{bits}b}"[::-1], 2)
    assert lines["ROW3"] == g["leaves"][rev].tolist()
    sib = np.array(lines["PROOF5"], dtype=np.uint64).reshape(-1, 4)
    assert oracle.merkle_verify(g["leaves"][5], 5, g["cap"], sib)
    assert lines["FFT_OF_COEFFS_EQUALS_VALUES"] == [1]
    assert "CAP_TOO_BIG error -1" in p.stdout
