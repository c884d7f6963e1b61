#!/usr/bin/env python3
"""One C5 shard (10 M x 768 fp16 rows), 256 queries per step through the pipelined session: a few steps, for a rocprofv3
kernel trace of what a step is made of (scripts/rocpd_gaps.py on the result)."""
import sys, time
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import oramacore_amd as oa  # noqa: E402
from oramacore_amd import _native as N  # noqa: E402
from oramacore_amd.shard_group import ShardGroup  # noqa: E402

group = ShardGroup([0])
ctx = group.ctx(0)
n, d, k, Q = 10_000_000, 768, 100, int(sys.argv[1]) if len(sys.argv) > 1 else 256
st = oa.EmbeddingFieldStorage(ctx, dimensions=d, reserve_rows=n, dtype=N.DTYPE_F16)
st.fill_synthetic(n, seed=0xC0FFEE)
qs = np.random.default_rng(1).standard_normal((Q * 4, d)).astype(np.float32)
sess = group.session([st], qs, Q, k, n_slots=1)
for i in range(3):
    sess.step(i)
sess.sync()
t0 = time.perf_counter()
for i in range(3, 11):
    sess.step(i)
sess.sync()
print(f"{(time.perf_counter() - t0) / 8 * 1e3:.3f} ms per step of {Q} queries")
