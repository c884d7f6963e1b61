#!/usr/bin/env python3
"""Single full-text calls back to back (C4-shaped queries): calls per second, and with ORAMA_POST_CALL_TRACE=1 the host's phases
(mean microseconds per phase over 2 000 calls, on stderr)."""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import oramacore_amd as oa  # noqa: E402
from oramacore_amd import fulltext as ft  # noqa: E402

n, T, k = 10_000_000, 12, 100
rng = np.random.default_rng(0xB26)
ranks = np.unique(np.exp(rng.uniform(np.log(100), np.log(100000), size=2048)).astype(np.uint32))
ctx = oa.Context(0)
post = ft.PostingsStore(ctx)
post.fill_synthetic(n, ranks, seed=0xB25)
refs = [[(t, int(l), 1.0) for t, l in enumerate(rng.choice(len(ranks), size=T, replace=False))] for _ in range(64)]
for r in refs[:8]:
    post.search(r, T, float(n), k)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
if len(sys.argv) > 2:  # "always" / "never": compact key lists for every batch size / for none (default: batches of 8 and more)
    ctx.set_bm25_ranges(True, True, compact_keys="always" if sys.argv[2] == "always" else False)
t0 = time.perf_counter()
for i in range(N):
    post.search(refs[i % 64], T, float(n), k)
el = time.perf_counter() - t0
print(f"{N / el:.0f} single calls per second ({el / N * 1e6:.1f} us per call)")
