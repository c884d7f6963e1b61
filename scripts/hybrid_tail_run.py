#!/usr/bin/env python3
"""C4-shaped one-call hybrid queries (10 M x 768 fp32 + 10 M-doc BM25F, 12 tokens) for a kernel trace of the tail:
   rocprofv3 --kernel-trace -d out -- python scripts/hybrid_tail_run.py ; python scripts/hybrid_tail_run.py --report out
The report lists, per query, every kernel that starts after the scan began: offset from the scan's END, duration."""
import sys
from pathlib import Path
import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def report(path):
    import sqlite3
    db = next(Path(path).rglob("*results.db"))
    con = sqlite3.connect(str(db))
    rows = con.execute("select start, end, stream_id, name from kernels order by start").fetchall()
    scans = [i for i, r in enumerate(rows) if "vec_scan_f32" in r[3] or "vec_scan_f16" in r[3]]
    if any("rerank" in r[3] for r in rows):  # two-stage: one fp16 scan per query, the re-rank follows
        scans = [i for i in scans if "f16" in rows[i][3]]
    for qi in scans[-3:]:
        s0, e0 = rows[qi][0], rows[qi][1]
        nxt = next((rows[j][0] for j in scans if rows[j][0] > s0), None)
        print(f"--- scan {(e0 - s0) / 1e3:.1f} us")
        for s, e, st, name in rows:
            if s < s0 - 200_000 or (nxt and s >= nxt) or s > e0 + 2_000_000:
                continue
            short = name.replace("(anonymous namespace)::", "").replace("void ", "").replace("orama::", "").split("(")[0][:60]
            print(f"   start {((s - e0) / 1e3):9.1f} us from scan end  +{(e - s) / 1e3:7.1f} us  s{st}  {short}")
        if nxt:
            print(f"   next scan starts {((nxt - e0) / 1e3):.1f} us after this scan's end")


if len(sys.argv) > 2 and sys.argv[1] == "--report":
    report(sys.argv[2])
    sys.exit(0)

import oramacore_amd as oa  # noqa: E402
from oramacore_amd import fulltext as ft  # noqa: E402

n, dim, k, T = 10_000_000, 768, 100, 12
ctx = oa.Context(0)
vec = oa.EmbeddingFieldStorage(ctx, dimensions=dim, reserve_rows=n, dtype=oa.DTYPE_F32_SHADOW16 if "--shadow" in sys.argv else oa.DTYPE_F32)
vec.fill_synthetic(n, seed=0xC0FFEE)
rng = np.random.default_rng(0xB26)
ranks = np.unique(np.exp(rng.uniform(np.log(100), np.log(100000), size=2048)).astype(np.uint32))
post = ft.PostingsStore(ctx)
post.fill_synthetic(n, ranks, seed=0xB25)
qv = np.random.default_rng(0xBEEF).standard_normal((12, dim)).astype(np.float32)
for i in range(12):
    refs = [(t, int(l), 1.0) for t, l in enumerate(rng.choice(len(ranks), size=T, replace=False))]
    post.hybrid_search(vec, qv[i], k, 0.0, refs, T, float(n), k)
ctx.synchronize()
