#!/usr/bin/env python3
"""The launches of the LAST call of a rocprofv3 kernel trace (rocpd .db), as a timeline: start offset, duration, gap to the
previous launch's end, (queue, stream), kernel.  A "call" = the launches behind the last gap of more than `--idle-us`.

    cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/t -o t -- python script.py ; python scripts/call_timeline.py /tmp/t/*_results.db
"""
import argparse
import re
import sqlite3

ap = argparse.ArgumentParser()
ap.add_argument("db")
ap.add_argument("--idle-us", type=float, default=300.0)
ap.add_argument("--calls", type=int, default=1, help="how many of the last calls to print")
ap.add_argument("--tail", type=int, default=0, help="print only the last N launches of each call")
ap.add_argument("--skip-tail", type=int, default=0, help="... ending this many launches before the call's end")
a = ap.parse_args()
db = sqlite3.connect(a.db)
cols = [c[1] for c in db.execute("pragma table_info('kernels')")]
qcol = "queue_id" if "queue_id" in cols else "0"
scol = "stream_id" if "stream_id" in cols else "0"
rows = db.execute(f"select start, end, {qcol}, {scol}, name from kernels order by start").fetchall()
short = lambda n: re.sub(r"\(anonymous namespace\)::|orama::|void ", "", n).split("(")[0][:64]
cuts = [0]
run_end = rows[0][1]
for i in range(1, len(rows)):
    if (rows[i][0] - run_end) / 1e3 > a.idle_us:
        cuts.append(i)
    run_end = max(run_end, rows[i][1])
cuts.append(len(rows))
for c in range(max(0, len(cuts) - 1 - a.calls), len(cuts) - 1):
    lo, hi = cuts[c], cuts[c + 1]
    t0 = rows[lo][0]
    end = max(r[1] for r in rows[lo:hi])
    busy = sum(r[1] - r[0] for r in rows[lo:hi]) / 1e3
    print(f"== call of {hi - lo} launches: {(end - t0) / 1e3:.1f} us from first start to last end, sum of kernel time {busy:.1f} us")
    prev_end = t0
    if a.tail:
        hi2 = hi - a.skip_tail
        lo = max(lo, hi2 - a.tail)
        hi = hi2
        prev_end = rows[lo][0]
    for s, e, q, st, n in rows[lo:hi]:
        print(f"  +{(s - t0) / 1e3:9.1f}  {(e - s) / 1e3:8.1f} us  gap {(s - prev_end) / 1e3:7.1f}  q{q} s{st}  {short(n)}")
        prev_end = max(prev_end, e)
