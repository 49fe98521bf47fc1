"""Committed golden vectors (tests/golden/*.npz, made by tests/golden/gen_golden.py from the independent
big-int model): the C oracle must reproduce them on the CPU, the HIP path on the GPU. Bit-exact."""
import glob
import os

import numpy as np
import pytest

from gpu_util import bitrev_perm, gpu  # noqa: F401

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NTT_FILES = sorted(glob.glob(os.path.join(GOLD, "ntt_*.npz")))
MERKLE_FILES = sorted(glob.glob(os.path.join(GOLD, "merkle_*.npz")))
COMMIT_FILE = os.path.join(GOLD, "commit_p135_2e6_r3_h4.npz")


def test_fixtures_present():
    assert len(NTT_FILES) == 3 and len(MERKLE_FILES) == 6 and os.path.exists(COMMIT_FILE)


# ------------------------------------------------------------------ oracle (CPU)
@pytest.mark.parametrize("path", NTT_FILES)
def test_oracle_ntt_golden(oracle, path):
    g = np.load(path)
    assert (oracle.canon(oracle.fft(g["x"])) == g["fft"]).all()
    assert (oracle.canon(oracle.ifft(g["x"])) == g["ifft"]).all()
    assert (oracle.canon(oracle.coset_lde(g["lde_coeffs"], 3)) == g["lde_rate8_natural"]).all()


@pytest.mark.parametrize("path", MERKLE_FILES)
def test_oracle_merkle_golden(oracle, path):
    g = np.load(path)
    dig, cap = oracle.merkle_tree(g["leaves"], int(g["cap_height"]))
    assert (oracle.canon(dig) == g["digests"]).all() and (oracle.canon(cap) == g["cap"]).all()


def test_oracle_commit_golden(oracle):
    g = np.load(COMMIT_FILE)
    r = oracle.commit_from_values(g["values"], int(g["rate_bits"]), int(g["cap_height"]), threads=2)
    for k in ("coeffs", "leaves", "digests", "cap"):
        assert (oracle.canon(r[k]) == g[k]).all(), k


# ------------------------------------------------------------------ HIP (GPU)
@pytest.mark.gpu
@pytest.mark.parametrize("path", NTT_FILES)
def test_hip_ntt_golden(gpu, path):
    import plonky2_gpu_amd as pg

    g = np.load(path)
    assert (pg.fft_with_options(gpu, g["x"]) == g["fft"]).all()
    assert (pg.ifft_with_options(gpu, g["x"]) == g["ifft"]).all()
    lde = pg.coset_lde_bit_reversed(gpu, g["lde_coeffs"], 3)
    assert (lde == g["lde_rate8_natural"][bitrev_perm(len(g["lde_rate8_natural"]).bit_length() - 1)]).all()


@pytest.mark.gpu
@pytest.mark.parametrize("path", MERKLE_FILES)
def test_hip_merkle_golden(gpu, path):
    import plonky2_gpu_amd as pg

    g = np.load(path)
    t = pg.MerkleTree.new(gpu, g["leaves"], int(g["cap_height"]))
    assert (t.cap == g["cap"]).all()
    assert t.digests.shape == g["digests"].shape and (t.digests == g["digests"]).all()


@pytest.mark.gpu
def test_hip_commit_golden(gpu):
    import plonky2_gpu_amd as pg

    g = np.load(COMMIT_FILE)
    b = pg.PolynomialBatch.from_values(gpu, g["values"], int(g["rate_bits"]), False, int(g["cap_height"]))
    assert (b.polynomials == g["coeffs"]).all()
    assert (b.merkle_tree.cap == g["cap"]).all()
    assert (b.merkle_tree.digests == g["digests"]).all()
    assert (b.merkle_tree.d_leaves.download().reshape(g["leaves"].shape) == g["leaves"]).all()
