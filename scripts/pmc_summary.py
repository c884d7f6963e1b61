#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV passes (one directory per pass) for one kernel into a JSON record.

HBM traffic follows MI355X_MICROARCH.md §HBM: FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
reports exactly 1/2 of the bytes of a wide (16 B/lane) coalesced streaming read, so it is doubled;
WRITE_SIZE is uncalibrated and taken as is.  Usage:
    python scripts/pmc_summary.py gpurun_out/pmc vec_scan_f32_kernel <alg_bytes_per_launch> [mean] > profiles/xxx.json
`mean`: launches of unequal size (K2's dense head + filter super-chunks) — use the per-launch MEAN instead of the
median; <alg_bytes_per_launch> is then the mean algorithmic bytes per launch as bench.py reports it.
"""
import csv
import glob
import json
import statistics
import sys


def main(root: str, kernel: str, alg_bytes: float, use_mean: bool = False) -> None:
    vals = {}
    dur = []
    for path in glob.glob(f"{root}/*/*counter_collection.csv"):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                if kernel not in row["Kernel_Name"]:
                    continue
                vals.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
                dur.append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    med = {k: (statistics.fmean(v) if use_mean else statistics.median(v)) for k, v in vals.items()}
    out = {"kernel": kernel, "launches_per_counter": {k: len(v) for k, v in vals.items()}, "statistic": "mean" if use_mean else "median", "median": med,
           "median_duration_us_under_pmc": statistics.median(dur) / 1e3 if dur else None}
    if "FETCH_SIZE" in med:
        fetch = med["FETCH_SIZE"] * 1024 * 2  # gfx950: x2 for wide coalesced streaming reads
        write = med.get("WRITE_SIZE", 0.0) * 1024
        out["hbm_read_bytes_corrected"] = fetch
        out["hbm_write_bytes_uncalibrated"] = write
        out["traffic_bytes_per_launch"] = fetch + write
        out["alg_bytes_per_launch"] = alg_bytes
        out["traffic_over_algorithmic"] = (fetch + write) / alg_bytes
    if "TCC_HIT_sum" in med and "TCC_MISS_sum" in med:
        out["l2_hit_rate"] = med["TCC_HIT_sum"] / (med["TCC_HIT_sum"] + med["TCC_MISS_sum"])
    if "SQ_WAVE_CYCLES" in med:
        wc = med["SQ_WAVE_CYCLES"]
        out["sq_fractions_of_wave_cycles"] = {k: med[k] / wc for k in
                                              ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY") if k in med}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], float(sys.argv[3]), len(sys.argv) > 4 and sys.argv[4] == "mean")
