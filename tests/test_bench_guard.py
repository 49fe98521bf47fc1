"""bench.py quotes counters (roofline.traffic, the Poseidon issue bound) from a committed summary that rocprofv3 produced in
separate runs. The summary records the sha256 of the kernel sources it was collected on; a summary of other sources must
not be quoted."""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _tree(tmp_path):
    root = tmp_path / "tree"
    for rel in bench.PMC_SOURCES:
        dst = root / rel
        dst.parent.mkdir(parents=True, exist_ok=True)
        shutil.copy(os.path.join(ROOT, rel), dst)
    (root / "profiles").mkdir()
    return root


def test_a_summary_of_these_sources_is_used(tmp_path):
    root = _tree(tmp_path)
    path = root / "profiles" / "x_pmc_summary.json"
    path.write_text(json.dumps({"source_sha256": bench.source_hashes(str(root)), "kernels": {}}))
    d, why = bench.pmc_summary(str(path), str(root))
    assert d is not None and why is None


def test_a_summary_of_other_sources_is_refused(tmp_path):
    root = _tree(tmp_path)
    path = root / "profiles" / "x_pmc_summary.json"
    path.write_text(json.dumps({"source_sha256": bench.source_hashes(str(root)), "kernels": {}}))
    with open(root / "plonky2_gpu_amd/csrc/ntt_direct.hip", "a") as f:
        f.write("// edited after the counters were collected\n")
    d, why = bench.pmc_summary(str(path), str(root))
    assert d is None and "ntt_direct.hip" in why and "regenerate" in why


def test_a_summary_without_hashes_or_a_missing_one_is_refused(tmp_path):
    root = _tree(tmp_path)
    path = root / "profiles" / "x_pmc_summary.json"
    path.write_text(json.dumps({"kernels": {}}))
    assert bench.pmc_summary(str(path), str(root))[0] is None
    assert bench.pmc_summary(str(root / "profiles" / "absent.json"), str(root))[0] is None


def test_the_committed_summary_if_any_matches_or_is_reported_stale():
    """Whatever state the tree is in, bench.py's answer is one of the two: counters of THIS tree's kernels, or null with a reason."""
    d, why = bench.pmc_summary()
    assert (d is None) != (why is None)
    traffic, note = bench.pmc_traffic(20, 64)
    assert (traffic is None) or traffic > 0
    assert note
