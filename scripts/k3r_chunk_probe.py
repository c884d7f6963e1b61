#!/usr/bin/env python3
"""C4-shaped BM25 batch of 32 queries (one set of launches): wall time and device time per query by kernel."""
import os, sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import oramacore_amd as oa
from oramacore_amd import fulltext as ft
n, T, k = 10_000_000, 12, 100
ctx = oa.Context(0)
if os.environ.get("K3R_FAST") is not None:  # (comparison flavour: run with ORAMA_COMPARISON_KERNELS=1)  # the plain batch's scoring body: 1 = bm25_ranges_fast.hip, 0 = the round-5 body
    ctx.set_option("k3r_fast", int(os.environ["K3R_FAST"]))
rng = np.random.default_rng(0xB26)
ranks = np.unique(np.exp(rng.uniform(np.log(100), np.log(100000), size=2048)).astype(np.uint32))
post = ft.PostingsStore(ctx)
post.fill_synthetic(n, ranks, seed=0xB25)
qs = [([(t, int(l), 1.0) for t, l in enumerate(rng.choice(len(ranks), size=T, replace=False))], T, None) for _ in range(64)]
for filtered in (False,):
    prep = post.prepare_batch(qs[:32], float(n), k)
    prep.run()
    ctx.prof_reset(); ctx.prof_enable(True)
    t0 = time.perf_counter()
    for _ in range(5):
        prep.run()
    el = (time.perf_counter() - t0) / 5
    ctx.prof_enable(False)
    print("filtered" if filtered else "classic", f"{el*1e6/32:.1f} us/query wall;", {kname: round(ctx.prof_get(kname)[0] * 1e3 / 5 / 32, 2) for kname in ("bm25_range_bounds", "bm25_range_score", "topk_select")}, flush=True)
