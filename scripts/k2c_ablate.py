#!/usr/bin/env python3
"""K2c ablations (timing only; ORAMA_K2C_DBG builds produce garbage results): where does the 5.2 ms of a
256-query pass over 10 M x 768 fp16 go?  One process, one corpus; the knob is read at every launch."""
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
NAMES = {0: "full kernel", 9: "DMA only (no LDS reads, no MFMA)", 13: "corpus DMA only (no query-fragment DMA, no compute)",
         1: "DMA + LDS fragment reads, no MFMA", 2: "LDS reads + MFMA + barriers, no DMA", 4: "corpus DMA + compute, no query-fragment DMA",
         11: "barriers + loop skeleton only"}
for dbg in [int(x) for x in os.environ.get("DBGS", "0,9,13,1,2,4,11,0").split(",")]:
    os.environ["ORAMA_K2C_DBG"] = str(dbg)
    for _ in range(2):
        st.storage_search(q, k)
    ctx.prof_reset(); ctx.prof_enable(True)
    reps = 4
    for _ in range(reps):
        st.storage_search(q, k)
    ctx.prof_enable(False)
    a = ctx.prof_get("vec_scan_f16")
    # the dense head (131072 rows) always runs the real kernel; subtract nothing, just report per pass
    print(f"K2C_DBG={dbg:2d} {NAMES.get(dbg, '?'):58s} scan {a[0]/reps:7.3f} ms per 256-query pass ({a[1]//reps} launches)", flush=True)
