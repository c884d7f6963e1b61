#!/usr/bin/env python3
"""Probe: K1m (fp32 MFMA batched scan on the plain fp32 store) — time per corpus pass, per call and queries/s for batches of
9..128 on 10M x 768 (the north-star rows), beside K1b (<= 8 queries per pass, VALU).  `--rows R --dim D` for other shapes."""
import argparse, sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import oramacore_amd as oa  # noqa: E402
import bench  # noqa: E402  (the clock / power sampler)

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=10_000_000)
ap.add_argument("--dim", type=int, default=768)
ap.add_argument("--batches", default="8,9,16,24,32,64,128")
ap.add_argument("--reps", type=int, default=8)
ap.add_argument("--head-rows", type=int, default=0, help="dense head of the filter pipeline (option f16_head_rows; 0 = 131 072)")
ap.add_argument("--plan", choices=["cvt", "mfma"], default="cvt", help="candidate scan of the batches: K1x (rows rounded to fp16 in registers) or K1m (f32 x f32)")
a = ap.parse_args()
ctx = oa.Context(0)
ctx.set_option("f32_batch_cvt", 1 if a.plan == "cvt" else 0)
if a.head_rows:
    ctx.set_option("f16_head_rows", a.head_rows)
try:
    bdf = ctx.pci_bus_id()
except Exception:  # noqa: BLE001
    bdf = None
n, d, k = a.rows, a.dim, 100
st = oa.EmbeddingFieldStorage(ctx, dimensions=d, reserve_rows=n)
st.fill_synthetic(n, seed=0xC0FFEE)
q = np.random.default_rng(1).standard_normal((256, d)).astype(np.float32)
alg = n * d * 4
for nq in [int(x) for x in a.batches.split(",")]:
    for _ in range(2):
        st.storage_search(q[:nq], k)
    ctx.prof_reset(); ctx.prof_enable(True)
    with bench.ClockSampler(bdf) as cs:
        t0 = time.perf_counter()
        for _ in range(a.reps):
            st.storage_search(q[:nq], k)
        el = (time.perf_counter() - t0) / a.reps * 1e3
    ctx.prof_enable(False)
    c = cs.summary()
    m = ctx.prof_get("vec_scan_f32_mfma"); x = ctx.prof_get("vec_scan_f32_cvt"); b = ctx.prof_get("vec_scan_f32_multi"); k1 = ctx.prof_get("vec_scan_f32"); s = ctx.prof_get("topk_select")
    per = 64 if x[1] else 32
    passes = -(-nq // per)
    if x[1]:
        m = x
    line = f"nq={nq:3d} call {el:8.3f} ms  QPS {nq/el*1e3:8.1f} | {'K1x' if x[1] else 'K1m'} {m[0]/a.reps:7.3f} ms in {m[1]//a.reps} launches"
    if m[1]:
        per_pass = m[0] / a.reps / passes
        line += f" = {per_pass:6.3f} ms per pass of <={per} ({alg/per_pass/1e6:7.1f} GB/s, {alg/per_pass/1e6/8000:5.3f} of HBM peak" + (")" if x[1] else f"; mfma {2*32*n*d/per_pass/1e9/157.3:5.3f} of 157.3 TF)")
    line += f" | K1b {b[0]/a.reps:7.3f} ms ({b[1]//a.reps}) | K1 {k1[0]/a.reps:7.3f} ms ({k1[1]//a.reps}) | select {s[0]/a.reps:6.3f} ms"
    if c.get("available"):
        line += f" | {c['sclk_mhz_median']:.0f} MHz, {c['power_w_mean']:.0f} W, PPT residency {c.get('ppt_throttle_residency_pct')}"
    print(line, flush=True)
st.close()
