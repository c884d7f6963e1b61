#!/bin/bash
# K2 ablations: which part keeps the fp16 scan at ~5.4 TB/s?
set +e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for D in 0 4 8 12; do
  echo "== ORAMA_F16_DEBUG=$D"
  ORAMA_F16_DEBUG=$D timeout 300 python - <<'PY'
import numpy as np, oramacore_amd as oa
ctx = oa.Context(0)
n, d, q = 10_000_000, 768, 64
st = oa.EmbeddingFieldStorage(ctx, dimensions=d, reserve_rows=n, dtype=oa.DTYPE_F16)
st.fill_synthetic(n, seed=1)
qs = np.random.default_rng(0).standard_normal((q, d)).astype(np.float32)
for kc, nb in ((8, 3), (12, 2)):
    ctx.set_f16_tuning(kc, nb)
    for _ in range(2): st.storage_search(qs, 100)
    ctx.prof_reset(); ctx.prof_enable(True)
    for _ in range(8): st.storage_search(qs, 100)
    ctx.prof_enable(False)
    ms, cnt = ctx.prof_get("vec_scan_f16")
    per = ms / 8
    print(f"kc={kc} nbuf={nb}: scan {per:.3f} ms/step  {n*768*2/per/1e6:.0f} GB/s")
PY
done
