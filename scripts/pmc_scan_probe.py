#!/usr/bin/env python3
"""A fixed number of session steps of one BASELINE vector configuration, for a `rocprofv3 --pmc` pass whose TOTAL traffic over
the scan kernels is divided by steps x algorithmic bytes (scripts/pmc_total.py): `python scripts/pmc_scan_probe.py c2|c3|c5 [steps]`.
(The fp16 scans run a dense head plus super-chunks of different sizes per step: per-launch figures mean little, the sum per
step is N x D x 2 bytes whatever the split.)"""
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import oramacore_amd as oa  # noqa: E402
from oramacore_amd.shard_group import ShardGroup  # noqa: E402

SHAPES = {"ns": (10_000_000, 768, 1, "f32"), "c2": (1_000_000, 384, 1, "f32"), "c3": (10_000_000, 768, 64, "f16"),
          "c5": (10_000_000, 768, 256, "f16"),
          # round 6: the plain fp32 store asked 64 queries per pass (K1x proposes; `nsb_mfma`: 32 per pass, K1m proposes)
          "nsb": (10_000_000, 768, 64, "f32"), "nsb_mfma": (10_000_000, 768, 32, "f32")}
name = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
n, d, qb, dt = SHAPES[name]
group = ShardGroup([0])
ctx = group.ctx(0)
if name == "nsb_mfma":
    ctx.set_option("f32_batch_cvt", 0)
st = oa.EmbeddingFieldStorage(ctx, dimensions=d, reserve_rows=n, dtype=oa.DTYPE_F16 if dt == "f16" else oa.DTYPE_F32)
st.fill_synthetic(n, seed=0xC0FFEE)
q = np.random.default_rng(0xBEEF).standard_normal((steps * qb, d)).astype(np.float32)
sess = group.session([st], q, qb, 100, n_slots=1 if dt == "f16" else 2)
for i in range(steps):
    sess.step(i)
sess.sync()
sess.close()
print(json.dumps({"workload": name, "steps": steps, "alg_bytes_per_step": n * d * (2 if dt == "f16" else 4), "queries_per_step": qb}))
