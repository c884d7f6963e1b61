#!/usr/bin/env python3
"""HIP-event spans of the three bracketed launch groups of a lone full-text call (bounds, score, top-k), mean over N calls."""
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
ctx = oa.Context(0)
post = ft.PostingsStore(ctx)
post.fill_synthetic(n, ranks, seed=0xB25)
refs = [[(t, int(l), 1.0) for t, l in enumerate(rng.choice(len(ranks), size=T, replace=False))] for _ in range(64)]
for r in refs[:8]:
    post.search(r, T, float(n), k)
N = 2000
ctx.prof_reset()
ctx.prof_enable(True)
t0 = time.perf_counter()
for i in range(N):
    post.search(refs[i % 64], T, float(n), k)
el = time.perf_counter() - t0
ctx.prof_enable(False)
out = {name: ctx.prof_get(name) for name in ("bm25_range_bounds", "bm25_range_score", "topk_select")}
print(f"{el / N * 1e6:.1f} us per call with the profiler on; spans (us): " + ", ".join(f"{k_} {v[0] / max(v[1], 1) * 1e3:.2f} x {v[1]}" for k_, v in out.items()))
