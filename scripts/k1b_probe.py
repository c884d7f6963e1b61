#!/usr/bin/env python3
"""Probe: K1b (fp32 multi-query scan) launch time and the select tail for batches of 1..16 on 10M x 768."""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import oramacore_amd as oa  # noqa: E402

ctx = oa.Context(0)
n, d, k = 10_000_000, 768, 100
st = oa.EmbeddingFieldStorage(ctx, dimensions=d, reserve_rows=n)
st.fill_synthetic(n, seed=0xC0FFEE)
q = np.random.default_rng(1).standard_normal((16, d)).astype(np.float32)
for nq in (1, 2, 4, 8, 16):
    for _ in range(2):
        st.storage_search(q[:nq], k)
    ctx.prof_reset(); ctx.prof_enable(True)
    t0 = time.perf_counter()
    reps = 10
    for _ in range(reps):
        st.storage_search(q[:nq], k)
    el = (time.perf_counter() - t0) / reps * 1e3
    ctx.prof_enable(False)
    a = ctx.prof_get("vec_scan_f32"); b = ctx.prof_get("vec_scan_f32_multi"); s = ctx.prof_get("topk_select")
    print(f"nq={nq:2d} call {el:7.3f} ms | K1 {a[0]/reps:6.3f} ms ({a[1]//reps} launches) | K1b {b[0]/reps:6.3f} ms ({b[1]//reps} launches) | select {s[0]/reps:6.3f} ms | QPS {nq/el*1e3:7.1f}", flush=True)
