#!/usr/bin/env python3
"""A lone full-text call with compact key lists (option k3r_compact 2: always) against the default (compact for batches of >= 8
queries, one slot per posting for a lone call): calls/s of C4-shaped queries and the device time per call by kernel."""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import oramacore_amd as oa  # noqa: E402
from oramacore_amd import fulltext as ft  # noqa: E402

n, T, k = 10_000_000, 12, 100
rng = np.random.default_rng(0xB26)
ranks = np.unique(np.exp(rng.uniform(np.log(100), np.log(100000), size=2048)).astype(np.uint32))
qs = [[(t, int(l), 1.0) for t, l in enumerate(rng.choice(len(ranks), size=T, replace=False))] for _ in range(400)]
out = {}
for mode in (1, 2):
    ctx = oa.Context(0)
    ctx.set_option("k3r_compact", mode)
    post = ft.PostingsStore(ctx)
    post.fill_synthetic(n, ranks, seed=0xB25)
    for q in qs[:50]:
        post.search(q, T, float(n), k)
    t0 = time.perf_counter()
    res = [post.search(q, T, float(n), k) for q in qs]
    el = time.perf_counter() - t0
    ctx.prof_reset(); ctx.prof_enable(True)
    for q in qs[:100]:
        post.search(q, T, float(n), k)
    ctx.prof_enable(False)
    dev = {kn: round(ctx.prof_get(kn)[0] * 1e3 / 100, 2) for kn in ("bm25_range_bounds", "bm25_range_score", "topk_select")}
    print(f"k3r_compact={mode}: {len(qs) / el:8.0f} lone calls/s ({el / len(qs) * 1e6:.1f} us per call); device us per call {dev}", flush=True)
    out[mode] = res
    post.close(); ctx.close()
same = all(a[2] == b[2] and np.array_equal(a[0], b[0]) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32)) for a, b in zip(out[1], out[2]))
print("identical answers:", same)
assert same
