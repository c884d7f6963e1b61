#!/usr/bin/env python3
"""K3r scoring launch, round 4 (sort-free: LDS bitmap -> document rank -> presence masks) against round 3 (merge tree of
64-bit keys), in ONE process of the comparison library (ORAMA_COMPARISON_KERNELS=1 -> liborama_hip_cmp.so; the second
context is created with ORAMA_K3R_MERGE=1).  Same synthetic C4 postings on both contexts, the same queries:
every answer must be bit-identical; device time per query by kernel from the library's HIP events, batch-entry rate.
Also: filtered batch (NOT-deleted bitmap), threshold, several lists per token (df counted on the device)."""
import os
import sys
import time
from pathlib import Path

import numpy as np

os.environ["ORAMA_COMPARISON_KERNELS"] = "1"
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import oramacore_amd as oa  # noqa: E402
from oramacore_amd import fulltext as ft  # noqa: E402

n, T, k = 10_000_000, 12, 100
rng = np.random.default_rng(0xB26)
ranks = np.unique(np.exp(rng.uniform(np.log(100), np.log(100000), size=2048)).astype(np.uint32))
os.environ.pop("ORAMA_K3R_MERGE", None)
ctx_new = oa.Context(0)
os.environ["ORAMA_K3R_MERGE"] = "1"
ctx_old = oa.Context(0)
os.environ.pop("ORAMA_K3R_MERGE", None)
posts = {}
for name, ctx in (("sort-free (r04)", ctx_new), ("merge tree (r03)", ctx_old)):
    p = ft.PostingsStore(ctx)
    p.fill_synthetic(n, ranks, seed=0xB25)
    posts[name] = (ctx, p)
NQ = 1024
plain = [([(t, int(l), 1.0) for t, l in enumerate(rng.choice(len(ranks), size=T, replace=False))], T, None) for _ in range(NQ)]
thr = [(q[0], T, 0.25) for q in plain[:256]]
# several lists per token: 6 tokens x 2..3 lists (prefix / fuzzy expansions, several fields): df is counted on the device
multi = []
for _ in range(256):
    ls = rng.choice(len(ranks), size=16, replace=False)
    refs, i = [], 0
    for t in range(6):
        for _r in range(2 + (t % 2)):
            refs.append((t, int(ls[i]), 1.0 + 0.5 * (i % 3)))
            i += 1
    multi.append((refs, 6, None))
allow_mask = (np.arange(n) % 7) != 3
bm = oa.AllowBitmap.from_mask(allow_mask)

results = {}
for name, (ctx, post) in posts.items():
    out = {}
    res_bm = bm.to_device(ctx)  # the same filter as a resident bitmap: document frequencies counted under it are remembered
    for tag, qs, allow in (("plain", plain, None), ("threshold", thr, None), ("multi-list", multi, None), ("filtered", plain[:256], bm),
                           ("filtered, resident bitmap", plain[:256], res_bm)):
        prep = post.prepare_batch(qs, float(n), k, allow=allow)
        t0 = time.perf_counter()
        prep.run()
        first = time.perf_counter() - t0
        if tag in ("multi-list", "filtered, resident bitmap"):
            print(f"{name:18s} {tag:10s} {len(qs) / first:10.0f} queries/s the FIRST time (document frequencies counted on the device)", flush=True)
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            prep.run()
        el = (time.perf_counter() - t0) / reps
        out[tag] = prep.results()
        print(f"{name:18s} {tag:10s} {len(qs) / el:10.0f} queries/s through orama_post_search_batch ({len(qs)} queries)", flush=True)
    # device time per query by kernel: chunks of 32 one at a time
    chunks = [post.prepare_batch(plain[i:i + 32], float(n), k) for i in range(0, 512, 32)]
    for c in chunks:
        c.run()
    ctx.prof_reset(); ctx.prof_enable(True)
    for c in chunks:
        c.run()
    ctx.prof_enable(False)
    dev = {kn: round(ctx.prof_get(kn)[0] * 1e3 / 512, 3) for kn in ("bm25_range_bounds", "bm25_range_df", "bm25_range_score", "topk_select")}
    print(f"{name:18s} device us/query by kernel: {dev}  total {sum(dev.values()):.2f}", flush=True)
    # single calls
    t0 = time.perf_counter()
    for q in plain[:200]:
        post.search(q[0], T, float(n), k)
    print(f"{name:18s} single calls: {200 / (time.perf_counter() - t0):8.0f} /s", flush=True)
    results[name] = out

a, b = results["sort-free (r04)"], results["merge tree (r03)"]
for tag in a:
    same = all(x[2] == y[2] and np.array_equal(x[0], y[0]) and np.array_equal(x[1].view(np.uint32), y[1].view(np.uint32)) for x, y in zip(a[tag], b[tag]))
    print(f"bit-identical answers [{tag}]: {same} ({len(a[tag])} queries)")
    assert same, tag
