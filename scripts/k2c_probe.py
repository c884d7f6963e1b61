#!/usr/bin/env python3
"""Probe: fp16 scan for wide batches (C5 per-GPU shape: 10M x 768 fp16, Q = 256)."""
import os, sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import oramacore_amd as oa  # noqa: E402
from oramacore_amd import _native as N  # noqa: E402

ctx = oa.Context(0)
n, d, k = 10_000_000, 768, 100
st = oa.EmbeddingFieldStorage(ctx, dimensions=d, reserve_rows=n, dtype=N.DTYPE_F16)
st.fill_synthetic(n, seed=0xC0FFEE)
q = np.random.default_rng(1).standard_normal((256, d)).astype(np.float32)
for nq in [int(x) for x in os.environ.get("NQ", "64,128,256").split(",")]:
    for _ in range(2):
        st.storage_search(q[:nq], k)
    ctx.prof_reset(); ctx.prof_enable(True)
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps):
        st.storage_search(q[:nq], k)
    el = (time.perf_counter() - t0) / reps * 1e3
    ctx.prof_enable(False)
    a = ctx.prof_get("vec_scan_f16"); s = ctx.prof_get("topk_select")
    flops = 2.0 * nq * n * d
    print(f"DBG={os.environ.get('ORAMA_K2C_DBG','0')} WIDE={os.environ.get('ORAMA_F16_WIDE','1')} nq={nq:3d} call {el:7.3f} ms | scan {a[0]/reps:7.3f} ms ({a[1]//reps} launches) "
          f"| select {s[0]/reps:6.3f} ms | QPS {nq/el*1e3:8.0f} | MFMA TF/s {flops/(a[0]/reps*1e-3)/1e12:6.0f}", flush=True)
