#!/usr/bin/env python3
"""VERDICT r04 next #4 (a threshold shared between the shards of a group), as a CEILING measurement: what would the fp16 filter
passes gain from the TIGHTEST threshold any exchange could hand them — the query's final k-th distance itself?

ORAMA_F16_TAU_ORACLE=1 makes every call run its filter passes under the final k-th distances the previous call left behind
(one ulp up); this script asks the same batch again and again, so that cap is exact and the answers are the product's.  A
shared threshold made of the shards' dense heads (k-th best of G x 131 072 rows) is looser than this oracle; whatever the
oracle does not gain, no exchange gains.  Shapes: the C5 per-GPU shard (10 M x 768 fp16, 256 queries) and C3 (64 queries).
Each arm in its own process (the switch is read once): python scripts/f16_tau_ceiling_probe.py"""
import os
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def arm(oracle: bool):
    import numpy as np

    import bench
    import oramacore_amd as oa

    ctx = oa.Context(0)
    bdf = ctx.pci_bus_id()
    n, d, k = 10_000_000, 768, 100
    st = oa.EmbeddingFieldStorage(ctx, dimensions=d, reserve_rows=n, dtype=oa.DTYPE_F16)
    st.fill_synthetic(n, seed=0xC0FFEE)
    rng = np.random.default_rng(0xBEEF)
    for q in (256, 64):
        qs = rng.standard_normal((q, d)).astype(np.float32)
        ref = st.storage_search(qs, k)
        for _ in range(8):  # every scratch set of the pool has answered this batch once
            st.storage_search(qs, k)
        ctx.prof_reset(); ctx.prof_enable(True)
        reps = 0
        with bench.ClockSampler(bdf) as clk:
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < 2.0:
                out = st.storage_search(qs, k)
                reps += 1
            el = time.perf_counter() - t0
        ctx.prof_enable(False)
        scan_ms, launches = ctx.prof_get("vec_scan_f16")
        sel_ms, _ = ctx.prof_get("topk_select")
        c = clk.summary()
        same = bool(np.array_equal(out[0], ref[0]) and np.array_equal(out[1].view(np.uint32), ref[1].view(np.uint32)))
        e = c.get("energy_j")
        print(f"{'oracle threshold' if oracle else 'product':17s} Q={q:3d}: scan {scan_ms / reps:6.3f} ms/pass ({launches / reps:.0f} launches) | selections "
              f"{sel_ms / reps:5.3f} ms | call {el / reps * 1e3:6.3f} ms | {c.get('sclk_mhz_median')} MHz, {c.get('power_w_from_energy_counter')} W, PPT "
              f"{c.get('ppt_throttle_residency_pct')} % | {e / reps if e else float('nan'):6.3f} J/call | answers equal to the first call's: {same}", flush=True)
    st.close()
    ctx.close()


if __name__ == "__main__":
    if len(sys.argv) > 1:
        arm(sys.argv[1] == "oracle")
    else:
        for name in ("product", "oracle", "product"):
            env = dict(os.environ, ORAMA_F16_TAU_ORACLE="1" if name == "oracle" else "0")
            subprocess.run([sys.executable, __file__, name], env=env, check=True)
