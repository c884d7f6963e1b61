#!/usr/bin/env python3
"""Single C4-shaped BM25 queries, one call at a time (orama_post_search): what a rocprofv3 kernel trace of this shows is the
chain a single query is — launches, copies and the gaps between them."""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import oramacore_amd as oa
from oramacore_amd import fulltext as ft
n, T, k = 10_000_000, 12, 100
ctx = oa.Context(0)
rng = np.random.default_rng(0xB26)
ranks = np.unique(np.exp(rng.uniform(np.log(100), np.log(100000), size=2048)).astype(np.uint32))
post = ft.PostingsStore(ctx)
post.fill_synthetic(n, ranks, seed=0xB25)
qs = [[(t, int(l), 1.0) for t, l in enumerate(rng.choice(len(ranks), size=T, replace=False))] for _ in range(60)]
for q in qs[:10]:
    post.search(q, T, float(n), k)
t0 = time.perf_counter()
for q in qs[10:]:
    post.search(q, T, float(n), k)
print(f"{(time.perf_counter() - t0) / 50 * 1e6:.1f} us per call")
