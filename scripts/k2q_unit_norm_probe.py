#!/usr/bin/env python3
"""VERDICT r04 next #5b, as a CEILING experiment with a kill criterion: would a unit-norm fp16 store type (rows stored as
fp16(x / |x|) + the fp32 norm beside them; the fast reject = 16 maxima against a per-lane bound, no norm reads from LDS, no
multiplies) take the 256-query pass of K2q under 4.0 ms?

The corpus is generated with rows of norm 1 (ORAMA_SYNTH_UNIT_NORM=1), where the ablation build ORAMA_K2C_DBG=64 — the fast
reject without the four LDS reads and sixteen multiplies per tile — answers (almost) like the product kernel, so its passing
rows, slow paths and appends are the real ones.  Same process, same store, same queries: product kernel, ablation, product
kernel again.  Reported: scan ms per 256-query pass (HIP events), the package's energy per pass and clock (bench.py's
sampler), agreement of the answers.  Keep the store type only if the pass drops below 4.0 ms; else the record is the result."""
import os
import sys
import time
from pathlib import Path

import numpy as np

os.environ["ORAMA_SYNTH_UNIT_NORM"] = "1"
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
import oramacore_amd as oa  # noqa: E402

ctx = oa.Context(0)
bdf = ctx.pci_bus_id()
n, d, k, q = 10_000_000, 768, 100, 256
st = oa.EmbeddingFieldStorage(ctx, dimensions=d, reserve_rows=n, dtype=oa.DTYPE_F16)
st.fill_synthetic(n, seed=0xC0FFEE)
qs = np.random.default_rng(0xBEEF).standard_normal((4 * q, d)).astype(np.float32)
answers = {}
for tag, dbg in (("product kernel", "0"), ("unit-norm fast reject (DBG 64)", "64"), ("product kernel again", "0")):
    os.environ["ORAMA_K2C_DBG"] = dbg
    for i in range(2):
        st.storage_search(qs[i * q:(i + 1) * q], k)
    ctx.prof_reset(); ctx.prof_enable(True)
    reps = 0
    with bench.ClockSampler(bdf) as clk:
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 2.0:
            out = st.storage_search(qs[(reps % 4) * q:(reps % 4 + 1) * q], k)
            reps += 1
        el = time.perf_counter() - t0
    ctx.prof_enable(False)
    scan_ms, launches = ctx.prof_get("vec_scan_f16")
    sel_ms, _ = ctx.prof_get("topk_select")
    c = clk.summary()
    answers[tag] = st.storage_search(qs[:q], k)
    e = c.get("energy_j")
    print(f"{tag:34s} scan {scan_ms / reps:6.3f} ms/pass ({launches / reps:.0f} launches; {n * 768 * 2 / (scan_ms / reps) / 1e6:6.0f} GB/s) | selections "
          f"{sel_ms / reps:5.3f} ms | call {el / reps * 1e3:6.3f} ms | sclk {c.get('sclk_mhz_median')} MHz, {c.get('power_w_from_energy_counter')} W, "
          f"PPT residency {c.get('ppt_throttle_residency_pct')} % | {e / reps if e else float('nan'):6.3f} J/call", flush=True)
a, b = answers["product kernel"], answers["unit-norm fast reject (DBG 64)"]
same_ids = float(np.mean(a[0] == b[0]))
print(f"answers of the ablation against the product kernel on the unit-norm corpus: {100 * same_ids:.3f} % of the ids equal, "
      f"max |distance difference| {float(np.max(np.abs(a[1] - b[1]))):.2e}")
