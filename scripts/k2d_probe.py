#!/usr/bin/env python3
"""K2c vs the K2d producer/consumer geometries on the C5 per-GPU shape (10 M x 768 fp16, 256 queries per pass):
scan time per pass from the library's HIP events + a bit-exact comparison of every geometry's result with K2c's."""
import os, sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import oramacore_amd as oa  # noqa: E402
from oramacore_amd import _native as N  # noqa: E402

ctx = oa.Context(0)
n, d, k = int(os.environ.get("ROWS", 10_000_000)), 768, 100
st = oa.EmbeddingFieldStorage(ctx, dimensions=d, reserve_rows=n, dtype=N.DTYPE_F16)
st.fill_synthetic(n, seed=(1 << 64) - 1 if os.environ.get("CONST_ROWS") else 0xC0FFEE)
q = np.random.default_rng(1).standard_normal((256, d)).astype(np.float32)
if os.environ.get("ZERO_Q"):
    q[:] = 0.0
    q[:, 0] = 1e-3
if os.environ.get("DBG"):
    os.environ["ORAMA_K2C_DBG"] = os.environ["DBG"]
ref = None
modes = [int(x) for x in os.environ.get("MODES", "1,2,3").split(",")]
for nq in [int(x) for x in os.environ.get("NQ", "256").split(",")]:
    for mode in modes:
        ctx.set_f16_wide(mode)
        try:
            for _ in range(2):
                ids, dist, cnt = st.storage_search(q[:nq], k)
        except Exception as e:  # noqa: BLE001
            print(f"mode {mode} nq {nq}: FAILED {e}", flush=True)
            continue
        ctx.prof_reset(); ctx.prof_enable(True)
        reps = 5
        t0 = time.perf_counter()
        for _ in range(reps):
            ids, dist, cnt = st.storage_search(q[:nq], k)
        el = (time.perf_counter() - t0) / reps * 1e3
        ctx.prof_enable(False)
        a = ctx.prof_get("vec_scan_f16"); s = ctx.prof_get("topk_select")
        if mode == modes[0]:
            ref = (ids.copy(), dist.copy())
        same = np.array_equal(ids, ref[0]) and np.array_equal(dist, ref[1])
        scan = a[0] / reps
        print(f"mode {mode} nq {nq:3d}: scan {scan:7.3f} ms/pass ({a[1]//reps} launches) = {n*768*2/scan/1e6:7.1f} GB/s corpus, "
              f"{2.0*nq*n*768/scan/1e9:6.0f} TF/s | call {el:7.3f} ms | select {s[0]/reps:6.3f} ms | == mode {modes[0]}: {same}", flush=True)
