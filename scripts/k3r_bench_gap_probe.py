#!/usr/bin/env python3
"""Why does bench.py's BM25 batch run at 177 K queries/s when scripts/k3r_repeat_probe.py reads 225-250 K on the same box?
One suspect per arm, same postings and queries."""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench
import oramacore_amd as oa
from oramacore_amd import fulltext as ft
from oramacore_amd.shard_group import ShardGroup
n, T, k = 10_000_000, 12, 100
arm = sys.argv[1]
rng = np.random.default_rng(0xB26)
ranks = np.unique(np.exp(rng.uniform(np.log(100), np.log(100000), size=2048)).astype(np.uint32))
group = None
if arm == "group_ctx":
    group = ShardGroup([0]); ctx = group.ctx(0)
else:
    ctx = oa.Context(0)
if arm == "sampler":
    with bench.ClockSampler(ctx.pci_bus_id()) as c:
        time.sleep(0.2)
vec = None
if arm == "vector_store":
    vec = oa.EmbeddingFieldStorage(ctx, dimensions=768, reserve_rows=n, dtype=oa.DTYPE_F32); vec.fill_synthetic(n, seed=1)
post = ft.PostingsStore(ctx)
post.fill_synthetic(n, ranks, seed=0xB25)
total = 25
qlists = [rng.choice(len(ranks), size=T, replace=False) for _ in range(total)]
refs = [[(t, int(l), 1.0) for t, l in enumerate(ql)] for ql in qlists]
if arm == "bench_queries" or arm == "hybrid_first":
    pass
if arm == "vec_big":
    big = oa.EmbeddingFieldStorage(ctx, dimensions=768, reserve_rows=1_000_000, dtype=oa.DTYPE_F32); big.fill_synthetic(1_000_000, seed=1)
    big.storage_search(np.ones(768, dtype=np.float32), 100)
elif arm.startswith("vec_search"):
    small = oa.EmbeddingFieldStorage(ctx, dimensions=64, dtype=oa.DTYPE_F32)
    small.insert_rows(np.arange(1000, dtype=np.uint64), np.random.default_rng(2).standard_normal((1000, 64)).astype(np.float32))
    for _ in range(int(arm[len("vec_search"):] or 1)):
        small.storage_search(np.ones(64, dtype=np.float32), 5)
if arm in ("hybrid_first", "hybrid_k3", "hybrid_one", "hybrid_limit0"):
    if arm == "hybrid_k3":
        ctx.set_bm25_ranges(True, hybrid=False)
    vec = oa.EmbeddingFieldStorage(ctx, dimensions=768, reserve_rows=1_000_000, dtype=oa.DTYPE_F32); vec.fill_synthetic(1_000_000, seed=1)
    qv = np.random.default_rng(1).standard_normal((total, 768)).astype(np.float32)
    calls = [post.prepare_hybrid(vec, qv[i], 0 if arm == "hybrid_limit0" else k, 0.0, refs[i], T, float(n), k) for i in range(total)]
    for c in (calls[:1] if arm in ("hybrid_one", "hybrid_limit0") else calls): c.run()
    ctx.set_bm25_ranges(True)
batch_q = [(refs[i], T, None) for i in range(5, total)] * 102
post.search_batch(batch_q[:64], float(n), k)
prep = post.prepare_batch(batch_q, float(n), k)
prep.run()
ts = []
for _ in range(40 if "--trace" in sys.argv else 5):
    t0 = time.perf_counter(); prep.run(); ts.append(time.perf_counter() - t0)
chunks = [post.prepare_batch(batch_q[i:i + 32], float(n), k) for i in range(0, 512, 32)]
for c in chunks: c.run()
ctx.prof_reset(); ctx.prof_enable(True)
t0 = time.perf_counter()
for c in chunks: c.run()
wall = (time.perf_counter() - t0) / 512 * 1e6
ctx.prof_enable(False)
dev = {kn: round(ctx.prof_get(kn)[0] * 1e3 / 512, 2) for kn in ("bm25_range_bounds", "bm25_range_score", "topk_select")}
print(f"{arm:14s}: {len(batch_q) / np.median(ts):9.0f} queries/s (best {len(batch_q) / min(ts):9.0f}) | chunk by chunk: wall {wall:.2f} us/query, device {dev}", flush=True)
