#!/usr/bin/env python3
"""Full-text batches over a shard group vs the single store (C4-shaped queries, 10 M documents): orama_post_search_batch,
orama_shard_post_search_batch over 4 co-located shards, and the staged sharded query one by one."""
import sys, time
sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parent.parent))
import numpy as np, oramacore_amd as oa
from oramacore_amd import fulltext as ft
from oramacore_amd.shard_group import ShardGroup
n, T, k, S = 10_000_000, 12, 100, 4
rng = np.random.default_rng(0xB26)
ranks = np.unique(np.exp(rng.uniform(np.log(100), np.log(100000), size=2048)).astype(np.uint32))
qs = [([(t, int(l), 1.0) for t, l in enumerate(rng.choice(len(ranks), size=T, replace=False))], T, None) for _ in range(512)]
ctx = oa.Context(0)
single = ft.PostingsStore(ctx); single.fill_synthetic(n, ranks, seed=0xB25)
prep = single.prepare_batch(qs, float(n), k); prep.run()
t0 = time.perf_counter(); prep.run(); el = time.perf_counter() - t0
print(f"single store : {len(qs)/el:9.0f} queries/s (orama_post_search_batch)", flush=True)
single.close()
g = ShardGroup([0] * S)
shards = []
for i in range(S):
    p = ft.PostingsStore(g.ctx(i)); p.fill_synthetic(n // S, ranks, seed=0xB25 + i, first_doc_id=i * (n // S)); shards.append(p)
g.post_search_batch(shards, qs, float(n), k)
t0 = time.perf_counter(); g.post_search_batch(shards, qs, float(n), k); el = time.perf_counter() - t0
print(f"{S}-shard group: {len(qs)/el:9.0f} queries/s (orama_shard_post_search_batch, Python marshalling included)", flush=True)
t0 = time.perf_counter()
for q in qs[:64]:
    g.post_search(shards, q[0], T, float(n), k)
el = time.perf_counter() - t0
print(f"{S}-shard group: {64/el:9.0f} queries/s one by one (orama_shard_post_search, the staged query)", flush=True)
