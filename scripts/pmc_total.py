#!/usr/bin/env python3
"""Total HBM traffic of every launch whose kernel name holds <kernel>, from rocprofv3 --pmc CSV passes (one directory per counter),
against steps x algorithmic bytes: `python scripts/pmc_total.py <root> <kernel> <alg_bytes_per_step> <steps>`.
FETCH_SIZE / WRITE_SIZE are KiB; gfx950: FETCH_SIZE x 2 for wide coalesced streaming reads (MI355X_MICROARCH.md, HBM section);
WRITE_SIZE uncalibrated, taken as is."""
import csv
import glob
import json
import sys

root, kernel, alg, steps = sys.argv[1], sys.argv[2], float(sys.argv[3]), int(sys.argv[4])
tot, cnt, names = {}, {}, set()
for path in glob.glob(f"{root}/*/*counter_collection.csv"):
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if kernel not in row["Kernel_Name"]:
                continue
            c = row["Counter_Name"]
            tot[c] = tot.get(c, 0.0) + float(row["Counter_Value"])
            cnt[c] = cnt.get(c, 0) + 1
            names.add(row["Kernel_Name"].split("(")[0][-60:])
fetch = tot.get("FETCH_SIZE", 0.0) * 1024 * 2
write = tot.get("WRITE_SIZE", 0.0) * 1024
print(json.dumps({"kernel_filter": kernel, "kernels": sorted(names), "launches": cnt, "steps": steps,
                  "hbm_read_bytes_corrected_per_step": fetch / steps, "hbm_write_bytes_uncalibrated_per_step": write / steps,
                  "traffic_bytes_per_step": (fetch + write) / steps, "alg_bytes_per_step": alg,
                  "traffic_over_algorithmic": (fetch + write) / steps / alg}, indent=1))
