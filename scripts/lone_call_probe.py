#!/usr/bin/env python3
"""Lone calls with pauses between them, for scripts/call_timeline.py: `python scripts/lone_call_probe.py two_stage|hybrid_shadow|hybrid|bm25`."""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import oramacore_amd as oa  # noqa: E402
from oramacore_amd import fulltext as ft  # noqa: E402

arm = sys.argv[1] if len(sys.argv) > 1 else "two_stage"
n, d, k = 10_000_000, 768, 100
ctx = oa.Context(0)
rng = np.random.default_rng(3)
q = rng.standard_normal((8, d)).astype(np.float32)
times = []
if arm in ("two_stage", "hybrid_shadow", "hybrid"):
    dtype = oa.DTYPE_F32 if arm == "hybrid" else oa.DTYPE_F32_SHADOW16
    st = oa.EmbeddingFieldStorage(ctx, dimensions=d, reserve_rows=n, dtype=dtype)
    st.fill_synthetic(n, seed=1)
if arm == "two_stage":
    for i in range(12):
        t0 = time.perf_counter(); st.storage_search(q[i % 8], k); times.append(time.perf_counter() - t0)
        time.sleep(0.002)
else:
    T = 12
    ranks = np.unique(np.exp(rng.uniform(np.log(100), np.log(100000), size=2048)).astype(np.uint32))
    post = ft.PostingsStore(ctx)
    post.fill_synthetic(n, ranks, seed=0xB25)
    refs = [[(t, int(l), 1.0) for t, l in enumerate(rng.choice(len(ranks), size=T, replace=False))] for _ in range(8)]
    if arm == "bm25":
        for i in range(12):
            t0 = time.perf_counter(); post.search(refs[i % 8], T, float(n), k); times.append(time.perf_counter() - t0)
            time.sleep(0.002)
    else:
        hs = [post.prepare_hybrid(st, q[i], 10, 0.0, refs[i], T, float(n), 10) for i in range(8)]
        for i in range(12):
            t0 = time.perf_counter(); hs[i % 8].run(); times.append(time.perf_counter() - t0)
            time.sleep(0.002)
print(arm, "host wall per call, ms:", [round(t * 1e3, 3) for t in times])
