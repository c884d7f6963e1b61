#!/usr/bin/env python3
"""One 256-query fp16 search: single store (10 M x 768) vs a co-located group of 4 shards of 2.5 M rows — wall time per
call and device time by kernel family (the contexts' HIP-event profiler)."""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import oramacore_amd as oa
from oramacore_amd import _native as N
from oramacore_amd.shard_group import ShardGroup

n, d, k, nq = 10_000_000, 768, 100, int(sys.argv[1]) if len(sys.argv) > 1 else 256
q = np.random.default_rng(1).standard_normal((nq, d)).astype(np.float32)
names = ("vec_scan_f16", "topk_select")

def report(tag, ctx, call, reps=6):
    call(); call()
    ctx.prof_reset(); ctx.prof_enable(True)
    t0 = time.perf_counter()
    for _ in range(reps):
        call()
    el = (time.perf_counter() - t0) / reps * 1e3
    ctx.prof_enable(False)
    print(tag, f"{el:.3f} ms/call", {nm: (round(ctx.prof_get(nm)[0] / reps, 3), ctx.prof_get(nm)[1] // reps) for nm in names}, flush=True)

ctx = oa.Context(0)
st = oa.EmbeddingFieldStorage(ctx, dimensions=d, reserve_rows=n, dtype=N.DTYPE_F16)
st.fill_synthetic(n, seed=5)
report("single store ", ctx, lambda: st.storage_search(q, k))
st.close()
g = ShardGroup([0, 0, 0, 0])
shards = []
for i in range(4):
    s = oa.EmbeddingFieldStorage(g.ctx(i), dimensions=d, reserve_rows=n // 4, dtype=N.DTYPE_F16)
    s.fill_synthetic(n // 4, seed=5 + i, first_doc_id=i * (n // 4))
    shards.append(s)
report("4-shard group", g.ctx(0), lambda: g.vec_search(shards, q, k))
