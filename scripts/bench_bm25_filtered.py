#!/usr/bin/env python3
"""BM25 batch throughput under the NOT-deleted filter (resident bitmap, 99 % allowed) vs unfiltered, C4 full-text shape."""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import oramacore_amd as oa  # noqa: E402
from oramacore_amd import fulltext as ft  # noqa: E402

n, T, k = 10_000_000, 12, 100
ctx = oa.Context(0)
rng = np.random.default_rng(0xB26)
ranks = np.unique(np.exp(rng.uniform(np.log(100), np.log(100000), size=2048)).astype(np.uint32))
post = ft.PostingsStore(ctx)
post.fill_synthetic(n, ranks, seed=0xB25)
qlists = [rng.choice(len(ranks), size=T, replace=False) for _ in range(512)]
batch = [([(t, int(l), 1.0) for t, l in enumerate(ql)], T, None) for ql in qlists] * 2
bm = oa.AllowBitmap.from_mask(rng.random(n) < 0.99).to_device(ctx)
for name, allow in (("unfiltered", None), ("filtered (99 % allowed, df counted on the device)", bm)):
    post.search_batch(batch[:64], float(n), k, allow=allow)
    ctx.prof_reset()
    ctx.prof_enable(True)
    t0 = time.perf_counter()
    post.search_batch(batch, float(n), k, allow=allow)
    el = time.perf_counter() - t0
    ctx.prof_enable(False)
    prof = {nm: round(ctx.prof_get(nm)[0] * 1e3 / len(batch), 2) for nm in ("bm25_range_bounds", "bm25_range_df", "bm25_range_score", "topk_select")}
    print(f"{name:50s}: {len(batch) / el:9.0f} queries/s | device us per query {prof}", flush=True)
