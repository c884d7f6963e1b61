#!/usr/bin/env python3
"""Per-stage timeline of K2d geometry 2 on block 0 (s_memtime stamps, ORAMA_K2C_DBG=16/20/25 builds)."""
import os, sys
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
buf = oa.DeviceBuffer(ctx, (8192 + 512) * 8)
os.environ["ORAMA_K2D_TRACE"] = hex(buf.ptr)
ctx.set_f16_wide(2)
for dbg in (16, 25):
    os.environ["ORAMA_K2C_DBG"] = str(dbg)
    st.storage_search(q, k)
    buf.upload(np.zeros(8192 + 512, dtype=np.uint64))
    st.storage_search(q, k)   # the LAST filter launch of this call leaves its stamps
    t = buf.download(np.uint64, 1024 * 8).reshape(1024, 8).astype(np.int64)
    g = slice(100, 900)
    lb, li, lw, cb, cc = t[g, 0], t[g, 1], t[g, 2], t[g, 4], t[g, 5]
    period = np.diff(cb)
    print(f"DBG {dbg}: stage period (consumer barrier to barrier): median {np.median(period):.0f} ticks, mean {period.mean():.0f}, p90 {np.percentile(period, 90):.0f}")
    print(f"   loader : issue {np.median(li - lb):.0f} | counted wait {np.median(lw - li):.0f} (mean {np.mean(lw - li):.0f}) | barrier wait (next) {np.median(lb[1:] - lw[:-1]):.0f}")
    print(f"   consumer: stage body {np.median(cc - cb):.0f} (mean {np.mean(cc - cb):.0f}) | barrier wait (next) {np.median(cb[1:] - cc[:-1]):.0f} (mean {np.mean(cb[1:] - cc[:-1]):.0f})")
    ep = t[g, 6]
    if (ep > 0).any():
        e = ep[ep > 0] - cc[ep > 0]
        print(f"   epilogue (every {int(np.median(np.diff(np.nonzero(ep > 0)[0]))) if (ep>0).sum()>1 else 0} stages): median {np.median(e):.0f} ticks")
    blk = buf.download(np.uint64, 512, offset=8192 * 8).reshape(256, 2).astype(np.int64)
    dur = blk[:, 1] - blk[:, 0]
    start = blk[:, 0] - blk[:, 0].min()
    end = blk[:, 1] - blk[:, 0].min()
    print(f"   per-block duration of the LAST launch (ticks): min {dur.min()} median {int(np.median(dur))} max {dur.max()} | start skew max {start.max()} | "
          f"makespan {end.max()} | by XCD (block % 8) median: {[int(np.median(dur[x::8])) for x in range(8)]}")
