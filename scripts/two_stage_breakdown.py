#!/usr/bin/env python3
"""Where the time of a two-stage query goes (NS corpus, q = 1 / 8 / 64): HIP-event time of the shadow scan and of the
selections against the wall time per call."""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import oramacore_amd as oa  # noqa: E402

ctx = oa.Context(0)
n, dim, k = 10_000_000, 768, 100
st = oa.EmbeddingFieldStorage(ctx, dimensions=dim, dtype=oa.DTYPE_F32_SHADOW16, reserve_rows=n)
st.fill_synthetic(n, seed=0x5EED)
rng = np.random.default_rng(1)
import os
for qb in [int(x) for x in os.environ.get("QB", "1,8,64,256").split(",")]:
    qs = rng.standard_normal((12, qb, dim)).astype(np.float32)
    st.storage_search(qs[0], k)
    ctx.prof_reset()
    ctx.prof_enable(True)
    t0 = time.perf_counter()
    for i in range(1, 11):
        st.storage_search(qs[i], k)
    el = (time.perf_counter() - t0) / 10 * 1e3
    ctx.prof_enable(False)
    sc_ms, sc_n = ctx.prof_get("vec_scan_f16")
    se_ms, se_n = ctx.prof_get("topk_select")
    print(f"q={qb:3d}: {el:6.3f} ms per call | shadow scan {sc_ms / 10:6.3f} ms in {sc_n / 10:.0f} launches "
          f"({15.36 / (sc_ms / 10):.2f} TB/s over 15.36 GB) | selections {se_ms / 10:6.3f} ms in {se_n / 10:.0f} launches", flush=True)
