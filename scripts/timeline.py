#!/usr/bin/env python3
"""Print a compact kernel timeline (start, duration, stream/queue, name) from a rocprofv3 rocpd sqlite trace."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
lo = int(sys.argv[2]) if len(sys.argv) > 2 else 0
n = int(sys.argv[3]) if len(sys.argv) > 3 else 60
rows = db.execute("select start, end, queue_id, stream_id, name from kernels order by start").fetchall()
t0 = rows[0][0]
for s, e, q, st, name in rows[lo:lo + n]:
    short = name.split("(")[0].split("::")[-1][:40]
    print(f"{(s - t0) / 1e3:10.1f} us  +{(e - s) / 1e3:8.1f}  q{q} s{st}  {short}")
