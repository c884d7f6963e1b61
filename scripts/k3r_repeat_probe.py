#!/usr/bin/env python3
"""bench.py's BM25 batch (2 048 queries made of 20 distinct ones) against a batch of 2 048 distinct queries, compact key lists
against round 4's: is the batch entry's rate a property of the queries' repetition?"""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import oramacore_amd as oa
from oramacore_amd import fulltext as ft
n, T, k = 10_000_000, 12, 100
rng = np.random.default_rng(0xB26)
ranks = np.unique(np.exp(rng.uniform(np.log(100), np.log(100000), size=2048)).astype(np.uint32))
for compact in (True, False):
    ctx = oa.Context(0)
    ctx.set_bm25_ranges(True, compact_keys=compact)
    post = ft.PostingsStore(ctx)
    post.fill_synthetic(n, ranks, seed=0xB25)
    r2 = np.random.default_rng(0xB26)
    distinct = [([(t, int(l), 1.0) for t, l in enumerate(r2.choice(len(ranks), size=T, replace=False))], T, None) for _ in range(2048)]
    for tag, qs in (("2048 distinct", distinct), ("20 distinct x 102", distinct[:20] * 102), ("1 query x 2048", distinct[:1] * 2048)):
        prep = post.prepare_batch(qs, float(n), k)
        prep.run()
        ts = []
        for _ in range(5):
            t0 = time.perf_counter(); prep.run(); ts.append(time.perf_counter() - t0)
        print(f"{'compact' if compact else 'slots  '} {tag:20s}: {len(qs) / np.median(ts):9.0f} queries/s (best {len(qs) / min(ts):9.0f})", flush=True)
    post.close(); ctx.close()
