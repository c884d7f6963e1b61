#!/usr/bin/env python3
"""NS session steps back to back: wall per step with the library's event profiler off and on (bench.py's timed region runs with
it on: two event records per scan launch), for 1, 2 and 3 tail slots."""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import oramacore_amd as oa  # noqa: E402
from oramacore_amd.shard_group import ShardGroup  # noqa: E402

n, d, k = 10_000_000, 768, 100
group = ShardGroup([0])
ctx = group.ctx(0)
st = oa.EmbeddingFieldStorage(ctx, dimensions=d, reserve_rows=n, dtype=oa.DTYPE_F32)
st.fill_synthetic(n, seed=0xC0FFEE)
q = np.random.default_rng(1).standard_normal((64, d)).astype(np.float32)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
for slots in (2, 1, 3):
    for prof in (False, True, False, True):
        sess = group.session([st], q, 1, k, n_slots=slots)
        for i in range(5):
            sess.step(i)
        sess.sync()
        ctx.prof_reset()
        ctx.prof_enable(prof)
        t0 = time.perf_counter()
        for i in range(5, 5 + steps):
            sess.step(i)
        sess.sync()
        el = time.perf_counter() - t0
        ctx.prof_enable(False)
        extra = ""
        if prof:
            ms, cnt = ctx.prof_get("vec_scan_f32")
            extra = f"  scan by events {ms / max(cnt, 1):.4f} ms x {cnt}"
        print(f"slots {slots} profiler {'on ' if prof else 'off'}: {el / steps * 1e3:.4f} ms per step = {steps / el:.2f} QPS{extra}", flush=True)
        sess.close()
