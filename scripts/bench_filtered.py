#!/usr/bin/env python3
"""Filtered vector search at NS size (the path every search takes while deletes are pending: the NOT-deleted bitmap over
DocumentIds, index/filter.rs:344-392): plain fp32 store and fp32 + fp16 shadow, resident bitmap, 1 / 64 queries."""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import oramacore_amd as oa  # noqa: E402

ctx = oa.Context(0)
n, dim, k = 10_000_000, 768, 100
rng = np.random.default_rng(3)
mask = rng.random(n) < 0.99
bm = oa.AllowBitmap.from_mask(mask).to_device(ctx)
for name, dt in (("f32", oa.DTYPE_F32), ("f32+shadow16", oa.DTYPE_F32_SHADOW16)):
    st = oa.EmbeddingFieldStorage(ctx, dimensions=dim, dtype=dt, reserve_rows=n)
    st.fill_synthetic(n, seed=0x5EED)
    for qb, reps in ((1, 20), (64, 4 if name == "f32" else 10)):
        qs = rng.standard_normal((reps + 1, qb, dim)).astype(np.float32)
        for allow in (None, bm):
            st.storage_search(qs[reps], k, allow)
            t0 = time.perf_counter()
            for i in range(reps):
                st.storage_search(qs[i], k, allow)
            el = (time.perf_counter() - t0) / reps
            print(f"{name:14s} batch {qb:3d} {'filtered (99 % allowed)' if allow is not None else 'unfiltered':24s}: {el * 1e3:8.3f} ms per call, {qb / el:9.1f} q/s", flush=True)
    st.close()
