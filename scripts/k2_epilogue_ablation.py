#!/usr/bin/env python3
"""K2 (fp16 store, 64 queries, k = 100, 10 M x 768): HIP-event time of the scan launches with and without the epilogue
(ORAMA_K2_DBG=1 skips it in the filter-mode launches: answers are wrong, only the time is of interest)."""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import oramacore_amd as oa  # noqa: E402

ctx = oa.Context(0)
n, dim, k = 10_000_000, 768, 100
st = oa.EmbeddingFieldStorage(ctx, dimensions=dim, dtype=oa.DTYPE_F16, reserve_rows=n)
st.fill_synthetic(n, seed=0x5EED)
rng = np.random.default_rng(1)
for qb in (64, 256):
    qs = rng.standard_normal((8, qb, dim)).astype(np.float32)
    st.storage_search(qs[0], k)
    ctx.prof_reset()
    ctx.prof_enable(True)
    t0 = time.perf_counter()
    for i in range(1, 7):
        st.storage_search(qs[i], k)
    el = (time.perf_counter() - t0) / 6 * 1e3
    ctx.prof_enable(False)
    sc_ms, sc_n = ctx.prof_get("vec_scan_f16")
    print(f"q={qb:3d}: {el:6.3f} ms per call | scan {sc_ms / 6:6.3f} ms in {sc_n / 6:.0f} launches ({15.36 / (sc_ms / 6):.2f} TB/s over 15.36 GB)", flush=True)
