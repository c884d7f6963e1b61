#!/usr/bin/env python3
"""Per-step timeline of K2q on block 0 (s_memtime stamps of waves 0 and 4 — the two waves of one SIMD, the leading and the
lagging one; ORAMA_K2C_DBG=16 / 48 builds)."""
import os, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import oramacore_amd as oa  # noqa: E402
from oramacore_amd import _native as N  # noqa: E402

ctx = oa.Context(0)
n, d, k = int(os.environ.get("ROWS", 10_000_000)), 768, 100
st = oa.EmbeddingFieldStorage(ctx, dimensions=d, reserve_rows=n, dtype=N.DTYPE_F16)
st.fill_synthetic(n, seed=0xC0FFEE)
q = np.random.default_rng(1).standard_normal((256, d)).astype(np.float32)
words = 1024 * 16
buf = oa.DeviceBuffer(ctx, words * 8)
os.environ["ORAMA_K2D_TRACE"] = hex(buf.ptr)
ctx.set_f16_wide(4)
for dbg in [int(x) for x in os.environ.get("DBGS", "16,48").split(",")]:
    os.environ["ORAMA_K2C_DBG"] = str(dbg)
    st.storage_search(q, k)
    buf.upload(np.zeros(words, dtype=np.uint64))
    st.storage_search(q, k)  # the LAST filter launch of this call leaves its stamps
    t = buf.download(np.uint64, words).reshape(1024, 2, 8).astype(np.int64)
    g = slice(100, 900)
    for wv, name in ((0, "wave 0 (leading)"), (1, "wave 4 (lagging)")):
        arrive, rel, dma, done, epi, waited = (t[g, wv, i] for i in (0, 1, 2, 3, 4, 5))
        period = np.diff(rel)
        print(f"DBG {dbg} {name}: step period median {np.median(period):.0f} mean {period.mean():.0f} p90 {np.percentile(period, 90):.0f} ticks")
        print(f"    vmcnt wait {np.median(waited - arrive):.0f} (mean {np.mean(waited - arrive):.0f}) | barrier wait {np.median(rel - waited):.0f} (mean {np.mean(rel - waited):.0f})"
              f" | release -> stage multiplied {np.median(done - rel):.0f} (mean {np.mean(done - rel):.0f}) | stage end -> next arrival {np.median(arrive[1:] - done[:-1]):.0f}")
        has_dma = dma > 0
        if has_dma.any():
            print(f"    DMA issued {np.median((dma - rel)[has_dma]):.0f} ticks after the release")
        has_epi = epi > 0
        if has_epi.any():
            e = (epi - rel)[has_epi]
            print(f"    epilogue steps ({has_epi.sum()}): release -> epilogue done median {np.median(e):.0f} mean {e.mean():.0f}; "
                  f"their step period median {np.median(period[has_epi[:-1]]):.0f} vs others {np.median(period[~has_epi[:-1]]):.0f}")
