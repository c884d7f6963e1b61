#!/usr/bin/env python3
"""Timeline of the last N kernels of a rocprofv3 rocpd trace: start offset, duration, idle gap before each kernel.
Usage: python scripts/rocpd_gaps.py <results.db> [last_n]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
last_n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
rows = db.execute("select name, start, end from kernels order by start").fetchall()[-last_n:]
t0 = rows[0][1]
prev_end = rows[0][1]
busy = 0
for name, start, end in rows:
    short = name.replace("void ", "").replace("(anonymous namespace)::", "").replace("orama::", "")
    short = short.split("<")[0].split("(")[0][:40]
    print(f"{(start - t0) / 1e3:10.1f} us  +{(end - start) / 1e3:8.1f} us  gap {max(0, start - prev_end) / 1e3:7.1f} us  {short}")
    busy += end - start
    prev_end = max(prev_end, end)
print(f"span {(prev_end - t0) / 1e6:.3f} ms, busy {busy / 1e6:.3f} ms")
