#!/usr/bin/env python3
"""A batch of 2 048 C4-shaped BM25 queries through orama_post_search_batch, 40 times: with ORAMA_POST_CALL_TRACE=1 the library
prints where the HOST spends each set of 32 queries (tables 7 us, enqueues 15 us, waiting for the device 340 us with two sets in
flight: the batch rate is the device's)."""
import sys, time
from pathlib import Path; sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import oramacore_amd as oa
from oramacore_amd import fulltext as ft
n, T, k = 10_000_000, 12, 100
ctx = oa.Context(0)
rng = np.random.default_rng(0xB26)
ranks = np.unique(np.exp(rng.uniform(np.log(100), np.log(100000), size=2048)).astype(np.uint32))
post = ft.PostingsStore(ctx)
post.fill_synthetic(n, ranks, seed=0xB25)
qs = [([(t, int(l), 1.0) for t, l in enumerate(rng.choice(len(ranks), size=T, replace=False))], T, None) for _ in range(2048)]
prep = post.prepare_batch(qs, float(n), k)
prep.run()
t0 = time.perf_counter()
for _ in range(40):
    prep.run()
print(f"{2048 * 40 / (time.perf_counter() - t0):.0f} queries/s")
