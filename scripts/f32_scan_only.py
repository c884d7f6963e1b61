#!/usr/bin/env python3
"""Run a few K1 scans (NS shape) — target for rocprofv3 --pmc passes."""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import oramacore_amd as oa  # noqa: E402

ctx = oa.Context(0)
n, d = 5_000_000, 768
st = oa.EmbeddingFieldStorage(ctx, dimensions=d, reserve_rows=n)
st.fill_synthetic(n, seed=1)
qs = np.random.default_rng(0).standard_normal((4, d)).astype(np.float32)
for i in range(4):
    st.storage_search(qs[i], 100)
