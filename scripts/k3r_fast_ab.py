#!/usr/bin/env python3
"""K3r's scoring launch for plain top-k batches: the round-6 body (bm25_ranges_fast.hip: singletons without ranks, lists under
the published floor not scored) against the round-5 body (bm25_ranges.hip), in ONE process of the comparison library on two
contexts (option k3r_fast 1 / 0).  Same synthetic C4 postings, the same queries: every answer must be bit-identical —
plain, threshold, several lists per token, filtered (host words and resident bitmap), k = 1 ... 300; device time per
query by kernel (HIP events), batch-entry rate, single calls."""
import os
import sys
import time
from pathlib import Path

import numpy as np

os.environ["ORAMA_COMPARISON_KERNELS"] = "1"  # the fast body is a comparison unit (liborama_hip_cmp.so)
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import oramacore_amd as oa  # noqa: E402
from oramacore_amd import fulltext as ft  # noqa: E402

n, T, k = 10_000_000, 12, 100
rng = np.random.default_rng(0xB26)
ranks = np.unique(np.exp(rng.uniform(np.log(100), np.log(100000), size=2048)).astype(np.uint32))
ctx_new, ctx_old = oa.Context(0), oa.Context(0)
ctx_old.set_option("k3r_fast", 0)
ctx_new.set_option("k3r_fast", 1)
ctx_new.set_option("bm25_dense_acc", 1)  # the store of this context keeps bitmaps of its longest lists (background lists)
if len(sys.argv) > 1:  # compact lists for every batch size (single calls included) on both
    ctx_old.set_option("k3r_compact", 2)
    ctx_new.set_option("k3r_compact", 2)
posts = {}
for name, ctx in (("fast body (r06)", ctx_new), ("round-5 body", ctx_old)):
    p = ft.PostingsStore(ctx)
    p.fill_synthetic(n, ranks, seed=0xB25)
    posts[name] = (ctx, p)
NQ = 1024
plain = [([(t, int(l), 1.0) for t, l in enumerate(rng.choice(len(ranks), size=T, replace=False))], T, None) for _ in range(NQ)]
thr = [(q[0], T, 0.25) for q in plain[:256]]
multi = []
for _ in range(256):
    ls = rng.choice(len(ranks), size=16, replace=False)
    refs, i = [], 0
    for t in range(6):
        for _r in range(2 + (t % 2)):
            refs.append((t, int(ls[i]), 1.0 + 0.5 * (i % 3)))
            i += 1
    multi.append((refs, 6, None))
# the same term in two "fields" (two lists that overlap heavily): ADVICE r04's cell-table case — lists 0/1 hold the same docs
allow_mask = (np.arange(n) % 7) != 3
bm = oa.AllowBitmap.from_mask(allow_mask)

results = {}
for name, (ctx, post) in posts.items():
    out = {}
    res_bm = bm.to_device(ctx)
    for tag, qs, allow, kk in (("plain", plain, None, k), ("threshold", thr, None, k), ("multi-list", multi, None, k),
                               ("filtered", plain[:256], bm, k), ("filtered, resident bitmap", plain[:256], res_bm, k),
                               ("k=1", plain[:128], None, 1), ("k=10", plain[:128], None, 10), ("k=300", plain[:128], None, 300)):
        prep = post.prepare_batch(qs, float(n), kk, allow=allow)
        prep.run()
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            prep.run()
        el = (time.perf_counter() - t0) / reps
        out[tag] = prep.results()
        print(f"{name:24s} {tag:28s} {len(qs) / el:10.0f} queries/s through orama_post_search_batch ({len(qs)} queries)", flush=True)
    chunks = [post.prepare_batch(plain[i:i + 32], float(n), k) for i in range(0, 512, 32)]
    for c in chunks:
        c.run()
    ctx.prof_reset(); ctx.prof_enable(True)
    for c in chunks:
        c.run()
    ctx.prof_enable(False)
    dev = {kn: round(ctx.prof_get(kn)[0] * 1e3 / 512, 3) for kn in ("bm25_range_bounds", "bm25_range_df", "bm25_range_score", "topk_select")}
    print(f"{name:24s} device us/query by kernel: {dev}  total {sum(dev.values()):.2f}", flush=True)
    t0 = time.perf_counter()
    for q in plain[:200]:
        post.search(q[0], T, float(n), k)
    print(f"{name:24s} single calls: {200 / (time.perf_counter() - t0):8.0f} /s", flush=True)
    results[name] = out

a, b = results["fast body (r06)"], results["round-5 body"]
for tag in a:
    same = all(x[2] == y[2] and np.array_equal(x[0], y[0]) and np.array_equal(x[1].view(np.uint32), y[1].view(np.uint32)) for x, y in zip(a[tag], b[tag]))
    print(f"bit-identical answers [{tag}]: {same} ({len(a[tag])} queries)")
    assert same, tag
