"""tools/reference_dumps.py against dump files in the reference's formats (circuit_builder.rs:1077-1186, oracle.rs:743-753,
prover.rs:829-877, cuda/test.cu:129-136, 412-428), written here by the CPU oracle playing the Rust side
(tests/reference_dump_writer.py): one proof of a circuit with the ed25519 shape and all 25 gate kinds in use."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def dumps(tmp_path_factory):
    import reference_dump_writer as w

    d = str(tmp_path_factory.mktemp("dumps"))
    w.write(d, degree_bits=6)
    return d


def test_dump_files_have_the_documented_shapes(dumps):
    n, n_ext = 64, 512
    size = lambda f: os.path.getsize(os.path.join(dumps, f)) // 8  # noqa: E731
    assert size("values.bin") == 234 * n and size("sigma_vecs.bin") == 80 * n and size("zs_partial_products.bin") == 20 * n
    for c, p in (("constants_sigmas_commitment", 88), ("zs_partial_products_commitment", 20), ("wires_commitment", 234)):
        assert size(c + ".polynomials.bin") == p * n and size(c + ".leaves.bin") == p * n_ext
        assert size(c + ".digests.bin") == 4 * 2 * (n_ext - 16) and size(c + ".caps.bin") == 4 * 16
    assert size("k_is.bin") == 80 and size("alphas.bin") == size("betas.bin") == size("gammas.bin") == 2
    assert size("quotient_values2.bin") == 2 * n_ext


@pytest.mark.gpu
@pytest.mark.parametrize("degree_bits", [10] + ([12] if os.environ.get("PLONKY2_SLOW_TESTS") else []))
def test_check_at_sizes_with_two_pass_transforms(tmp_path, degree_bits):
    """The same route at 2^10 rows (LDE 2^13: two-pass transforms and LDEs, the pipelined commit, several selector groups'
    worth of rows per gate kind), and at 2^12 when PLONKY2_SLOW_TESTS is set (the oracle needs about five minutes for that
    proof; tools/gpu_runs/slow_parity.sh runs it and profiles/ keeps the report). The oracle's Poseidon / NTT / Merkle
    primitives come from the C restatement (accel.c_backend), which the reference's known answers pin."""
    import reference_dump_writer as w

    d = str(tmp_path / "dumps")
    w.write(d, degree_bits=degree_bits)
    tool = os.path.join(ROOT, "tools", "reference_dumps.py")
    p = subprocess.run([sys.executable, tool, "check", d], capture_output=True, text=True, timeout=1800)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    rep = json.loads(p.stdout[p.stdout.index("{"):])
    assert rep["ok"] is True and rep["degree_bits"] == degree_bits
    assert len([k for k, v in rep.items() if isinstance(v, str) and v.startswith("equal")]) >= 14, rep
    assert rep["5 proof.bin"].startswith("equal")


@pytest.mark.gpu
def test_check_reproduces_every_cpu_dump_on_the_device(dumps):
    tool = os.path.join(ROOT, "tools", "reference_dumps.py")
    p = subprocess.run([sys.executable, tool, "check", dumps], capture_output=True, text=True, timeout=1200)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    rep = json.loads(p.stdout[p.stdout.index("{"):])
    assert rep["ok"] is True and rep["degree_bits"] == 6
    compared = [k for k, v in rep.items() if isinstance(v, str) and v.startswith("equal")]
    assert len(compared) >= 4 + 3 + 4 + 1 + 2, rep  # wires x4, constants_sigmas x3, zs x4, quotient, cap + proof
    assert rep["5 proof.bin"].startswith("equal")
    # a dump that does not match is reported, with the position of the first difference
    q = np.fromfile(os.path.join(dumps, "quotient_values2.bin"), dtype="<u8")
    q[5] ^= np.uint64(1)
    q.tofile(os.path.join(dumps, "quotient_values2.bin"))
    p = subprocess.run([sys.executable, tool, "check", dumps], capture_output=True, text=True, timeout=1200)
    assert p.returncode == 1
    rep = json.loads(p.stdout[p.stdout.index("{"):])
    assert "MISMATCH at element 5" in rep["4 quotient_values2 (quotient polynomial coefficients)"]
