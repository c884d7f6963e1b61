#!/usr/bin/env python3
"""From a rocprofv3 rocpd trace of scripts/k3r_bench_gap_probe2.py: the two batch phases (before / after the hybrid call) —
per kernel: launches, mean duration, and how much of the phase two streams ran kernels at the same time."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cols = [c[1] for c in db.execute("pragma table_info('kernels')")]
qcol = "queue_id" if "queue_id" in cols else "0"
scol = "stream_id" if "stream_id" in cols else "0"
rows = db.execute(f"select start, end, {qcol}, {scol}, name from kernels order by start").fetchall()
import re
names = [re.sub(r"\(anonymous namespace\)::|orama::|void ", "", r[4]).split("(")[0].split("<")[0] for r in rows]
# the hybrid call is the only launch of range_score_docs_kernel
# the hybrid call sits between the two batch phases: the first launch that is not a K3r batch kernel after the first compact launch
first_batch = next(i for i, nme in enumerate(names) if "range_score_compact" in nme)
batchy = lambda nme: any(x in nme for x in ("range_bounds", "range_score_compact", "keys_reduce", "keys_final", "copyBuffer", "fillBuffer"))
cut = next(i for i in range(first_batch, len(names)) if not batchy(names[i]))
end_h = next(i for i in range(cut, len(names)) if "range_score_compact" in names[i])
print("== the hybrid call's launches:")
for i in range(cut, end_h):
    print(f"   +{(rows[i][1] - rows[i][0]) / 1e3:8.1f} us  q{rows[i][2]} s{rows[i][3]}  {names[i][:60]}")
def phase(lo, hi, tag):
    sel = [r for r, nme in zip(rows[lo:hi], names[lo:hi]) if "range_" in nme or "keys_" in nme]
    if not sel: return
    # keep the last 5 batch runs' worth: everything (simple)
    by = {}
    for (s, e, q, st, _), nme in zip(rows[lo:hi], names[lo:hi]):
        if "range_" in nme or "keys_" in nme:
            d = by.setdefault(nme[:34], [0, 0.0, set()]); d[0] += 1; d[1] += (e - s) / 1e3; d[2].add((q, st))
    t0, t1 = min(r[0] for r in sel), max(r[1] for r in sel)
    busy = sum((r[1] - r[0]) for r in sel) / 1e3
    print(f"== {tag}: {len(sel)} launches over {(t1 - t0) / 1e3:.0f} us, sum of kernel durations {busy:.0f} us (ratio {busy / ((t1 - t0) / 1e3):.2f})")
    per_stream = {}
    for (s_, e_, q_, st_, _), nme in zip(rows[lo:hi], names[lo:hi]):
        if "range_score_compact" in nme:
            per_stream[(q_, st_)] = per_stream.get((q_, st_), 0) + 1
    print(f"   scoring launches per (queue, stream): {per_stream}")
    cps = {}
    for (s_, e_, q_, st_, _), nme in zip(rows[lo:hi], names[lo:hi]):
        if "copyBuffer" in nme or "fillBuffer" in nme:
            cps[(nme[:24], q_, st_)] = cps.get((nme[:24], q_, st_), 0) + 1
    print(f"   copy / fill kernels per (name, queue, stream): {cps}")
    # timeline of the first 24 scoring launches of the phase
    sc_rows = [(r[0], r[1], r[2], r[3]) for r, nme in zip(rows[lo:hi], names[lo:hi]) if "range_score_compact" in nme][40:64]
    t00 = sc_rows[0][0]
    print("   scoring launches 40..63: " + " ".join(f"[s{st} {int((s0 - t00) / 1e3)}+{int((e0 - s0) / 1e3)}]" for s0, e0, q0, st in sc_rows))
    for k, (c, tot, qs) in sorted(by.items()):
        print(f"   {k:36s} {c:6d} launches, mean {tot / c:8.2f} us, (queue, stream) {sorted(qs)}")
phase(first_batch, cut, "before the hybrid call")
phase(end_h, len(rows), "after the hybrid call")
