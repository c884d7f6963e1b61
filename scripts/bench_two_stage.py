#!/usr/bin/env python3
"""Two-stage exact search (fp32 rows + fp16 shadow) against the plain fp32 scan on the north-star corpus
(10 M x 768, cosine top-100), through the host-buffer API (orama_vec_search: H2D queries, D2H results, one call per
batch).  Every answer of the shadow store is compared with the plain store's — ids and distance bits."""
import argparse, json, sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import oramacore_amd as oa  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=10_000_000)
ap.add_argument("--dim", type=int, default=768)
ap.add_argument("--k", type=int, default=100)
ap.add_argument("--batches", default="1,8,64,256")
ap.add_argument("--reps", type=int, default=20)
args = ap.parse_args()
ctx = oa.Context(0)
n, dim, k = args.rows, args.dim, args.k
stores = {}
for name, dt in (("f32", oa.DTYPE_F32), ("f32+shadow16", oa.DTYPE_F32_SHADOW16)):
    st = oa.EmbeddingFieldStorage(ctx, dimensions=dim, dtype=dt, reserve_rows=n)
    st.fill_synthetic(n, seed=0x5EED)
    stores[name] = st
rng = np.random.default_rng(0xBEEF)
out = {"metric": f"queries/s, exact cosine top-{k}, {n} x {dim} fp32 rows, host-buffer API", "by_batch": {}}
for qb in [int(x) for x in args.batches.split(",")]:
    reps = max(3, args.reps // max(1, qb // 8)) if qb > 1 else args.reps
    queries = rng.standard_normal((reps + 1, qb, dim)).astype(np.float32)
    row = {}
    answers = {}
    for name, st in stores.items():
        if name == "f32" and qb > 8:
            r = max(1, reps // 4)  # the plain store runs such a batch as qb / 8 corpus passes
        else:
            r = reps
        st.storage_search(queries[reps], k)
        t0 = time.perf_counter()
        for i in range(r):
            ans = st.storage_search(queries[i], k)
        el = time.perf_counter() - t0
        answers[name] = st.storage_search(queries[0], k)
        row[name] = {"qps": r * qb / el, "ms_per_batch": el / r * 1e3}
    a, b = answers["f32"], answers["f32+shadow16"]
    identical = bool(np.array_equal(a[0], b[0]) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32)) and np.array_equal(a[2], b[2]))
    assert identical, f"two-stage answer differs from the fp32 scan at batch {qb}"
    row["identical_to_fp32_scan"] = identical
    row["speedup"] = row["f32+shadow16"]["qps"] / row["f32"]["qps"]
    out["by_batch"][qb] = row
    print(f"batch {qb:4d}: fp32 scan {row['f32']['qps']:9.1f} q/s ({row['f32']['ms_per_batch']:8.2f} ms) | two-stage "
          f"{row['f32+shadow16']['qps']:9.1f} q/s ({row['f32+shadow16']['ms_per_batch']:8.2f} ms) | x{row['speedup']:.2f} | identical",
          flush=True)
info = stores["f32+shadow16"].info()
out["two_stage_queries"] = info["two_stage_queries"]
out["two_stage_fallbacks"] = info["two_stage_fallbacks"]
out["hbm_bytes"] = {kname: st.info()["hbm_bytes"] for kname, st in stores.items()}
print(json.dumps(out))
