#!/usr/bin/env python3
"""a query whose lists overlap heavily — the same term in two fields, terms that occur together — holds
more multi-posting documents per range than the scoring launch's cell tables take (512 documents / 1 024 cells): the range
overflows, the query is scored once for nothing and rerun with narrower ranges.  Round 5 remembers the width that held per
list set (orama_post::shrink_hint) and makes the first shrink step 4x instead of 8x.  This script times exactly that case:
1 M documents, list A in field 0 and list B in field 1 over THE SAME 400 K documents, plus two unrelated lists; batches of
256 queries [token 0 -> A and B, token 1 -> C]: the first call (every query pays the wasted pass) against the calls after it,
beside the same batch over NON-overlapping lists of the same lengths."""
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import oramacore_amd as oa  # noqa: E402
from oramacore_amd import fulltext as ft  # noqa: E402

n = 1_000_000
rng = np.random.default_rng(5)
docs = np.arange(n, dtype=np.uint64)
lens = rng.integers(4, 200, size=n)


def plist(field, idx):
    return ft.PostingList(field=field, docs=docs[idx], tf=rng.integers(1, 4, size=len(idx)), field_len=lens[idx])


same = np.sort(rng.choice(n, size=400_000, replace=False))
other = [np.sort(rng.choice(n, size=400_000, replace=False)) for _ in range(2)]
third = np.sort(rng.choice(n, size=100_000, replace=False))
ctx = oa.Context(0)
post = ft.PostingsStore(ctx)
post.build(docs, [float(lens.mean()), float(lens.mean())], [plist(0, same), plist(1, same), plist(0, other[0]), plist(1, other[1]), plist(0, third)])
overlap = [([(0, 0, 1.0), (0, 1, 2.0), (1, 4, 1.0)], 2, None)] * 256
disjoint = [([(0, 2, 1.0), (0, 3, 2.0), (1, 4, 1.0)], 2, None)] * 256
for tag, qs in (("lists over the SAME documents (token 0 in two fields)", overlap), ("lists of the same lengths over independent documents", disjoint)):
    rates = []
    for call in range(4):
        t0 = time.perf_counter()
        res = post.search_batch(qs, float(n), 100)
        rates.append(len(qs) / (time.perf_counter() - t0))
    print(f"{tag:58s}: call 1 {rates[0]:8.0f} queries/s, calls 2-4 {np.mean(rates[1:]):8.0f} (count {res[0][2]}, top-1 {int(res[0][0][0])})", flush=True)
one = overlap[0]
t0 = time.perf_counter()
for _ in range(50):
    post.search(one[0], 2, float(n), 100)
print(f"single calls over the overlapping lists: {50 / (time.perf_counter() - t0):.0f} /s")
