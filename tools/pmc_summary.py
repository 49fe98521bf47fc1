#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `bench.py --no-commit --no-cpu`
into profiles/<tag>_pmc_traffic.json (per-launch HBM-side bytes of the two NTT pass kernels).

Usage: python tools/pmc_summary.py <fetch_dir> <write_dir> <tag>
Corrections (MI355X_MICROARCH.md §HBM): counters are in KiB; on gfx950 FETCH_SIZE tallies the
128-B requests of a wide coalesced 16 B/lane stream at 64 B, i.e. reads exactly half — that
applies to the row pass (whole contiguous rows), not to the column pass whose 64-B segments are
single 64-B requests (calibrated: the column pass reads 128 MiB per launch and the counter says
130 MiB including twiddle tables). WRITE_SIZE is exact for 16 B/lane stores.
"""
import collections
import csv
import glob
import json
import sys


def collect(d, counter):
    agg = collections.defaultdict(list)
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            name = r["Kernel_Name"]
            key = "pass_A<10,true>" if "<10, true>" in name else "pass_B<10,false>" if "<10, false>" in name else name[:60]
            agg[(key, int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
    return agg


def med(v):
    v = sorted(v)
    return v[len(v) // 2]


def main():
    fetch_dir, write_dir, tag = sys.argv[1:4]
    fe, wr = collect(fetch_dir, "FETCH_SIZE"), collect(write_dir, "WRITE_SIZE")
    out = {"unit": "bytes per launch (16-column chunk of 2^20-point columns = 128 MiB of field elements)", "kernels": {}}
    total = 0.0
    for key in sorted(fe):
        k, grid = key
        f_kib, w_kib = med(fe[key]), min(wr.get(key, [0.0]))  # min = forward launches (aligned stores)
        corr = 2.0 if "pass_B" in k else 1.0
        rd, wrb = f_kib * 1024 * corr, w_kib * 1024
        out["kernels"][f"{k} grid={grid}"] = {
            "FETCH_SIZE_KiB_median": f_kib, "fetch_correction": corr, "read_bytes": rd,
            "WRITE_SIZE_KiB_forward": w_kib, "WRITE_SIZE_KiB_median_all": med(wr.get(key, [0.0])), "write_bytes": wrb,
            "launches": len(fe[key]),
        }
        if grid == 1048576:
            total += rd + wrb
    out["forward_chunk_total_bytes"] = total
    out["algorithmic_bytes_per_chunk"] = 16.0 * (1 << 20) * 16
    out["traffic_over_algorithmic"] = total / out["algorithmic_bytes_per_chunk"]
    path = f"profiles/{tag}_pmc_traffic.json"
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
