#!/usr/bin/env python3
"""Summarise the rocprofv3 passes of tools/gpu_runs/pmc_passes.sh (one directory per counter set, each a run of
`bench.py --no-prove --no-cpu` or of tools/bench_poseidon.py) into profiles/<tag>_pmc_summary.json.

Usage: python tools/pmc_summary.py <gpurun_out/r02pmc> <tag>

Per kernel and grid size: the median of every counter over the launches, the average duration from the un-instrumented
--kernel-trace --stats pass, and derived figures. Unit notes (MI355X_MICROARCH.md): SQ_WAVE_CYCLES / SQ_WAIT_* /
SQ_ACTIVE_INST_* count quad-cycles; FETCH_SIZE / WRITE_SIZE are KiB; on gfx950 FETCH_SIZE tallies the 128-byte requests of
a 16 B/lane streaming read at 64 bytes, i.e. reports half the bytes — the x2 correction is applied to EVERY kernel here
(all global reads of these kernels are 16 B/lane), and the copy kernel of the same run (a known byte count) is listed as
the calibration."""
import collections
import csv
import glob
import json
import os
import sys

SIMDS = 1024  # 256 CUs x 4


def short(name):
    for key in ("ntt_col_direct_kernel", "ntt_row_natural_direct_kernel", "ntt_row_inplace_direct_kernel", "ntt_pass_wave_kernel", "ntt_pass_kernel", "hash_leaves_chunk_kernel", "hash_leaves_kernel", "gate_constraints_kernel", "tree_layer_coop_kernel", "tree_layer_kernel",
                "transpose_kernel", "permute_batch_kernel", "copy16_kernel", "coset_tables_kernel"):
        if key in name:
            if "ntt_" in name and "<" in name:
                args = name[name.index("<") + 1:name.index(">")].replace(" ", "")
                return f"{key}<{args}>"
            return key
    return name[:48]


def med(v):
    v = sorted(v)
    return v[len(v) // 2]


def main():
    root, tag = sys.argv[1], sys.argv[2]
    counters = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(root, "*", "*", "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            key = (short(r["Kernel_Name"]), int(r["Grid_Size"]))
            counters[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur = collections.defaultdict(list)
    for f in glob.glob(os.path.join(root, "stats_*", "*", "*kernel_trace.csv")):
        for r in csv.DictReader(open(f)):
            dur[(short(r["Kernel_Name"]), int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, repo)
    import bench  # the list of kernel sources and their hashes live with the reader of this file (bench.py pmc_summary)

    out = {"units": "counter medians per launch; durations in microseconds from the un-instrumented kernel trace",
           "source_sha256": bench.source_hashes(repo),
           "source_sha256_note": "sha256 of the kernel sources these counters were collected on; bench.py quotes the counters only "
                                 "while the tree's files still hash to these values",
           "ntt_columns_per_launch": 64,
           "kernels": {}}
    for key in sorted(counters):
        name, grid = key
        c = {k: med(v) for k, v in counters[key].items()}
        e = {"grid_threads": grid, "launches_in_counter_passes": max(len(v) for v in counters[key].values()), "counters": c}
        if dur.get(key):
            e["avg_us"] = sum(dur[key]) / len(dur[key]) / 1e3
            e["timed_launches"] = len(dur[key])
        d = {}
        if c.get("SQ_WAVES"):
            d["valu_insts_per_wave"] = c.get("SQ_INSTS_VALU", 0) / c["SQ_WAVES"]
        if c.get("SQ_WAVE_CYCLES"):
            d["wave_parked_frac (s_waitcnt / barrier)"] = c.get("SQ_WAIT_ANY", 0) / c["SQ_WAVE_CYCLES"]
            d["wave_issue_stall_frac"] = c.get("SQ_WAIT_INST_ANY", 0) / c["SQ_WAVE_CYCLES"]
            d["wave_active_frac"] = c.get("SQ_ACTIVE_INST_ANY", 0) / c["SQ_WAVE_CYCLES"]
        if c.get("GRBM_GUI_ACTIVE") and c.get("SQ_INSTS_VALU"):
            cyc = c["GRBM_GUI_ACTIVE"] / 8.0  # summed over the 8 XCDs
            d["kernel_cycles"] = cyc
            d["valu_issue_frac_at_4_cycles_per_inst (lower bound of VALU busy)"] = c["SQ_INSTS_VALU"] * 4.0 / (SIMDS * cyc)
        if c.get("SQ_INSTS_MFMA") and c.get("SQ_WAVES"):
            d["matrix_insts_per_wave"] = c["SQ_INSTS_MFMA"] / c["SQ_WAVES"]
            if c.get("SQ_VALU_MFMA_BUSY_CYCLES") and c.get("GRBM_GUI_ACTIVE"):
                d["matrix_pipe_busy_frac (SQ_VALU_MFMA_BUSY_CYCLES / SIMDs / kernel cycles)"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (SIMDS * c["GRBM_GUI_ACTIVE"] / 8.0)
        if c.get("SQ_LDS_IDX_ACTIVE"):
            d["lds_bank_conflict_frac"] = c.get("SQ_LDS_BANK_CONFLICT", 0) / c["SQ_LDS_IDX_ACTIVE"]
        if "FETCH_SIZE" in c:
            d["read_bytes (FETCH_SIZE KiB x 1024 x 2)"] = c["FETCH_SIZE"] * 1024 * 2
            d["read_bytes_uncorrected"] = c["FETCH_SIZE"] * 1024
        if "WRITE_SIZE" in c:
            d["write_bytes (WRITE_SIZE KiB x 1024)"] = c["WRITE_SIZE"] * 1024
        if c.get("TCC_HIT_sum") is not None and c.get("TCC_MISS_sum"):
            d["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
        e["derived"] = d
        out["kernels"][f"{name} grid={grid}"] = e
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", f"{tag}_pmc_summary.json")
    json.dump(out, open(path, "w"), indent=1)
    # brief table
    for k, e in out["kernels"].items():
        d = e["derived"]
        print(f"{k[:64]:64s} us={e.get('avg_us', 0):9.1f} valu/wave={d.get('valu_insts_per_wave', 0):8.0f} "
              f"parked={d.get('wave_parked_frac (s_waitcnt / barrier)', 0):.2f} valu>={d.get('valu_issue_frac_at_4_cycles_per_inst (lower bound of VALU busy)', 0):.2f} "
              f"ldsconf={d.get('lds_bank_conflict_frac', 0):.2f} rd={d.get('read_bytes (FETCH_SIZE KiB x 1024 x 2)', 0) / 2**20:8.1f}MiB wr={d.get('write_bytes (WRITE_SIZE KiB x 1024)', 0) / 2**20:8.1f}MiB")


if __name__ == "__main__":
    main()
