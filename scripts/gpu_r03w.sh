#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03w
mkdir -p $O
python - <<'PY' 2>&1 | tee gpurun_out/r03w/score_map_memory.log
import time, numpy as np, oramacore_amd as oa
from oramacore_amd import fulltext as ft
n, T = 10_000_000, 12
ctx = oa.Context(0)
rng = np.random.default_rng(0xB26)
ranks = np.unique(np.exp(rng.uniform(np.log(100), np.log(100000), size=2048)).astype(np.uint32))
post = ft.PostingsStore(ctx); post.fill_synthetic(n, ranks, seed=0xB25)
refs = [(t, int(l), 1.0) for t, l in enumerate(rng.choice(len(ranks), size=T, replace=False))]
import ctypes
hip = ctypes.CDLL('libamdhip64.so')
def free_bytes():
    f, t = ctypes.c_size_t(), ctypes.c_size_t()
    assert hip.hipMemGetInfo(ctypes.byref(f), ctypes.byref(t)) == 0
    return f.value
for on in (True, False):
    ctx.set_bm25_ranges(on)
    free0 = free_bytes()
    sm = post.search_scores(refs, T, float(n), 100); sm.close()
    t0 = time.perf_counter()
    for _ in range(5):
        sm = post.search_scores(refs, T, float(n), 100); cnt = len(sm); sm.close()
    el = (time.perf_counter() - t0) / 5
    free1 = free_bytes()
    print(("K3r" if on else "K3 "), f"score map of {cnt} documents: {el*1e3:.3f} ms per search_scores call, scratch kept by the pool afterwards: {(free0-free1)/1e6:.0f} MB", flush=True)
PY
