#!/usr/bin/env python3
"""Why do two profiles of the same north-star kernel disagree by 3.4 %?

One lease, one process: the NS scan (10 M x 768 fp32, one query per step through the pipelined session) is timed
  A. cold    — the first thing the process does after filling the store,
  B. steady  — again after 300 more steps,
  C. after a power-hungry neighbour — 4 s of the fp16 256-query scan (K2q, the package at its power limit) on another store,
  D. recovered — after 2 s of idling,
each with the sampler's record of the SAME region: shader clock, MEMORY clock (uclk), fabric clock, socket power, PPT residency,
HBM / hotspot temperature, memory-controller activity.  One table row per phase; run on several leases and compare
(profiles/r06_ns_spread_*.log)."""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import oramacore_amd as oa  # noqa: E402
import bench  # noqa: E402
from oramacore_amd.shard_group import ShardGroup  # noqa: E402

group = ShardGroup([0])
ctx = group.ctx(0)
bdf = ctx.pci_bus_id()
n, d, k = 10_000_000, 768, 100
st = oa.EmbeddingFieldStorage(ctx, dimensions=d, reserve_rows=n)
st.fill_synthetic(n, seed=0xC0FFEE)
qs = np.random.default_rng(0xBEEF).standard_normal((64, d)).astype(np.float32)
sess = group.session([st], qs, 1, k, n_slots=2)


def phase(name, steps=100):
    ctx.prof_reset(); ctx.prof_enable(True)
    with bench.ClockSampler(bdf) as cs:
        t0 = time.perf_counter()
        for i in range(steps):
            sess.step(i % 64)
        sess.sync()
        wall = (time.perf_counter() - t0) / steps * 1e3
    ctx.prof_enable(False)
    s = np.asarray(ctx.prof_samples("vec_scan_f32"), dtype=np.float64)
    c = cs.summary()
    g = lambda key, fmt="{:.0f}": "n/a" if c.get(key) is None else fmt.format(c[key])  # noqa: E731
    print(f"| {name:28s} | {np.median(s):.4f} | {np.percentile(s, 5):.4f} .. {np.percentile(s, 95):.4f} | {wall:.4f} | {n * d * 4 / np.median(s) / 1e6:.0f} | "
          f"{g('sclk_mhz_median')} | {g('uclk_mhz_median')} ({g('uclk_mhz_min')} min) | {g('socclk_mhz_median')} | {g('power_w_mean')} / {g('power_w_from_energy_counter')} | "
          f"{g('ppt_throttle_residency_pct', '{:.0f} %')} | {g('temp_hbm_c_max')} / {g('temp_hotspot_c_max')} | {g('umc_activity_pct_median')} | {g('xcd_busy_spread_pct', '{:.1f}')} |", flush=True)


print("| phase | scan ms (median) | p05 .. p95 | step wall ms | GB/s | gfxclk MHz | uclk MHz | socclk MHz | W (samples / energy counter) | PPT residency | HBM / hotspot C | UMC activity % | XCD busy spread % |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|---|")
sess.step(0); sess.sync()
phase("A cold (first 100 steps)")
for i in range(300):
    sess.step(i % 64)
sess.sync()
phase("B steady (after 400 steps)")
hot = oa.EmbeddingFieldStorage(ctx, dimensions=d, reserve_rows=n, dtype=oa.DTYPE_F16)
hot.fill_synthetic(n, seed=0xC0FFEE)
q256 = np.random.default_rng(2).standard_normal((256, d)).astype(np.float32)
t0 = time.perf_counter()
with bench.ClockSampler(bdf) as cs:
    while time.perf_counter() - t0 < 4.0:
        hot.storage_search(q256, k)
c = cs.summary()
print(f"| (neighbour: fp16 256-query scans, 4 s) | | | | | {c.get('sclk_mhz_median')} | {c.get('uclk_mhz_median')} | {c.get('socclk_mhz_median')} | {c.get('power_w_mean')} | {c.get('ppt_throttle_residency_pct')} | {c.get('temp_hbm_c_max')} / {c.get('temp_hotspot_c_max')} | {c.get('umc_activity_pct_median')} | |")
phase("C right after the neighbour")
time.sleep(2.0)
phase("D after 2 s idle")
phase("E steady again")
sess.close(); hot.close(); st.close(); group.close()
