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


def test_traffic_counts_exactly_the_column_and_the_row_pass():
    """a renamed template argument dropped the column pass from the sum unnoticed. On the committed summaries the two
    passes of the forward transform are found by template name + pinned arguments at the batch's grid, and their HBM-side bytes
    are twice the algorithmic bytes (the second pass re-reads and re-writes everything)."""
    import glob

    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0[3-9]_pmc_summary.json")))
    assert paths
    for path in paths:
        d = json.load(open(path))
        c, r, why = bench.forward_ntt_kernels(d)
        assert why is None, (path, why)
        assert c[1].startswith("ntt_col_direct_kernel<") and r[1].startswith("ntt_row_natural_direct_kernel<false")
        assert c[0] == r[0] == max(bench._kernel_key(n)[2] for n in d["kernels"] if n.startswith("ntt_col_direct_kernel"))
        total, why = bench.traffic_of_summary(d, 64)
        assert why is None
        assert 1.9 <= total / (16.0 * (1 << 20) * 64) <= 2.2, (path, total)
        alu = bench.int_alu_of_summary(d)
        assert alu and 0.3 < alu["frac"] <= 1.0


def test_a_summary_with_a_missing_or_duplicated_pass_is_an_error_not_a_partial_sum():
    d = json.load(open(os.path.join(ROOT, "profiles", "r03_pmc_summary.json")))
    only_row = {"kernels": {k: v for k, v in d["kernels"].items() if not k.startswith("ntt_col_direct_kernel")}}
    assert bench.traffic_of_summary(only_row, 64)[0] is None and "column" in bench.traffic_of_summary(only_row, 64)[1]
    renamed = {"kernels": {k.replace("ntt_col_direct_kernel<2,true,false>", "ntt_col_direct_kernel<2,true>"): v for k, v in d["kernels"].items()}}
    assert bench.traffic_of_summary(renamed, 64)[0] is None
    twice = {"kernels": dict(d["kernels"])}
    twice["kernels"]["ntt_col_direct_kernel<1,true,false> grid=262144"] = d["kernels"]["ntt_col_direct_kernel<2,true,false> grid=262144"]
    assert bench.traffic_of_summary(twice, 64)[0] is None


def test_the_committed_bench_line_agrees_with_itself_and_with_the_committed_kernel_trace():
    """Three stopwatches on one quantity, the duration of a batch transform (one launch pair): the roofline's own (HIP events around
    the timed region / launch pairs), the throughput (`value` = transforms per second), and rocprofv3's kernel durations of the same
    steady command (profiles/r05_ntt_kernel_stats.csv, collected under the profiler on the same device in the same gpurun call,
    tools/gpu_runs/r05_pmc_and_bench.sh). Until the end of round 4 the roofline used per-pair events and read 3-6 % above the other two."""
    import csv
    for tag, col_name in (("r04", "ntt_col_direct_kernel<2, true, false>"), ("r05", "ntt_col_direct_kernel<2, true, false, false>"),
                          ("r06", "ntt_col_direct_kernel<2, true, false, false, false>")):
        line = json.loads(open(os.path.join(ROOT, "profiles", tag + "_bench.json")).read())
        r = line["roofline"]
        assert "timed region" in r["ms_definition"]
        per_batch_ms = 1e3 * line["config"]["batch_columns"] / line["value"]
        assert abs(r["ms"] - per_batch_ms) / per_batch_ms < 0.01, (r["ms"], per_batch_ms)
        assert abs(r["frac"] - 16.0 * (1 << line["config"]["log_n"]) * line["config"]["batch_columns"] / (r["ms"] * 1e-3) / 8e12) < 1e-6
        col = row = None
        for k in csv.DictReader(open(os.path.join(ROOT, "profiles", tag + "_ntt_kernel_stats.csv"))):
            if col_name + "(" in k["Name"]:
                col = float(k["AverageNs"]) * 1e-6
            if "ntt_row_natural_direct_kernel<false>" in k["Name"]:
                row = float(k["AverageNs"]) * 1e-6
        assert col and row, tag
        assert abs((col + row) - r["ms"]) / r["ms"] < 0.03, (tag, col, row, r["ms"])   # the profiler costs a per cent or two
        assert r["ms_forward_pairs_bracketed_one_by_one"]["median"] >= r["ms"]


def test_the_documents_quote_the_committed_bench_line():
    """DESIGN.md / README.md quoted a commit time that was not in the artifact they cited. The figures of the current
    round's tables are the ones of profiles/r06_bench.json (collected on one device in one call together with the kernel statistics
    and counters), to the precision they are printed with. Since round 6 the line carries the whole metric as top-level scalars — the
    LAST keys of the line — and they agree with the objects they summarise."""
    line = json.loads(open(os.path.join(ROOT, "profiles", "r06_bench.json")).read())
    e = line["extra"]
    want = ["%.1f k NTT/s" % (line["value"] / 1e3), "%.3f" % line["roofline"]["frac"], "%.1f ms" % e["commit_ms"], "%.1f ms" % e["prove"]["prove_ms"],
            "%.0f M leaves/s" % (e["merkle_leaves_per_s"] / 1e6), "%.1f proofs/s" % e["prove_in_flight"]["proofs_per_s"]]
    tail = ["ntts_per_s", "ntt_hbm_frac", "prove_ms", "prove_proofs_per_s", "prove_proofs_per_s_in_flight", "prove_in_flight", "prove_wires_commitment_ms",
            "prove_quotient_polys_ms", "commit_ms", "merkle_leaves_per_s", "commit_hbm_frac", "commit_cpu_baseline_ms", "prove_cpu_baseline_ms"]
    assert list(line)[-len(tail):] == tail, list(line)[-len(tail):]
    assert line["prove_ms"] == e["prove"]["prove_ms"] and line["commit_ms"] == e["commit_ms"] and line["merkle_leaves_per_s"] == e["merkle_leaves_per_s"]
    assert line["prove_proofs_per_s_in_flight"] == e["prove_in_flight"]["proofs_per_s"] > line["prove_proofs_per_s"]
    assert line["roofline"]["prove_ms"] == line["prove_ms"] and line["cpu_baseline"]["prove_ms"] == line["prove_cpu_baseline_ms"]
    assert len(json.dumps({k: line[k] for k in tail})) < 1200  # they fit a 2 000-character tail with room to spare
    for doc in ("DESIGN.md", "README.md"):
        text = open(os.path.join(ROOT, doc)).read()
        for w in want:
            assert w in text, (doc, w)
    # the traffic and issue figures come from the counter summary of the same call, which bench.py accepted (source hashes equal)
    assert line["roofline"]["traffic"] and abs(line["roofline"]["traffic_over_algorithmic"] - 2.0) < 0.1
