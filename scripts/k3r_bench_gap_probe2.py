#!/usr/bin/env python3
"""Sequence: plain batch (fast?) -> one hybrid call -> the same batch (slow?) -> sleep -> batch -> K3 hybrid -> batch."""
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
ctx = oa.Context(0)
post = ft.PostingsStore(ctx)
post.fill_synthetic(n, ranks, seed=0xB25)
qlists = [rng.choice(len(ranks), size=T, replace=False) for _ in range(25)]
refs = [[(t, int(l), 1.0) for t, l in enumerate(ql)] for ql in qlists]
vec = oa.EmbeddingFieldStorage(ctx, dimensions=768, reserve_rows=1_000_000, dtype=oa.DTYPE_F32); vec.fill_synthetic(1_000_000, seed=1)
qv = np.random.default_rng(1).standard_normal((25, 768)).astype(np.float32)
batch_q = [(refs[i], T, None) for i in range(5, 25)] * 102
prep = post.prepare_batch(batch_q, float(n), k)
def rate(tag):
    prep.run()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); prep.run(); ts.append(time.perf_counter() - t0)
    print(f"{tag:46s}: {len(batch_q) / np.median(ts):9.0f} queries/s", flush=True)
rate("batch, fresh process")
arm = sys.argv[1] if len(sys.argv) > 1 else "hybrid"
if arm == "k201":
    post.search(refs[0], T, float(n), 201)
    rate("batch after ONE plain search with top_k 201")
    post.search(refs[0], T, float(n), 300)
    rate("batch after ONE plain search with top_k 300")
elif arm == "small_hybrid":
    h = post.prepare_hybrid(vec, qv[0], 10, 0.0, refs[0], T, float(n), 10)
    h.run()
    rate("batch after ONE hybrid call, k = limit = 10")
elif arm.startswith("hy:"):
    _, kk, ll = arm.split(":")
    h = post.prepare_hybrid(vec, qv[0], int(ll), 0.0, refs[0], T, float(n), int(kk))
    ctx.prof_reset(); ctx.prof_enable(True)
    h.run()
    ctx.prof_enable(False)
    print("   the hybrid call launched:", {kn: ctx.prof_get(kn)[1] for kn in ("bm25_accumulate", "bm25_finalize", "bm25_range_score", "bm25_range_bounds", "topk_select", "vec_scan_f32")}, flush=True)
    rate(f"batch after ONE hybrid call, top_k {kk}, limit {ll}")
elif arm == "other_ctx":
    h = post.prepare_hybrid(vec, qv[0], 10, 0.0, refs[0], T, float(n), 100)
    h.run()
    rate("batch after ONE hybrid call (100, 10), same context")
    ctx2 = oa.Context(0)
    post2 = ft.PostingsStore(ctx2)
    post2.fill_synthetic(n, ranks, seed=0xB25)
    prep2 = post2.prepare_batch(batch_q, float(n), k)
    prep2.run()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); prep2.run(); ts.append(time.perf_counter() - t0)
    print(f"{'the same batch on a NEW context + store':46s}: {len(batch_q) / np.median(ts):9.0f} queries/s", flush=True)
    rate("batch on the first context again")
elif arm == "new_sets":
    import threading
    h = post.prepare_hybrid(vec, qv[0], 10, 0.0, refs[0], T, float(n), 100)
    h.run()
    rate("batch after ONE hybrid call (100, 10)")
    preps = [post.prepare_batch(batch_q, float(n), k) for _ in range(2)]
    th = [threading.Thread(target=pp.run) for pp in preps]
    for t in th: t.start()
    for t in th: t.join()
    rate("... after two batches at once (two more scratch sets exist)")
    rate("... again")
elif arm == "single100":
    post.search(refs[0], T, float(n), 100)
    rate("batch after ONE plain single search, top_k 100")
