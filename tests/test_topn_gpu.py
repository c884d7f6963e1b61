"""GPU parity of K4 (top_n, src/collection_manager/sides/read/sort.rs:260-279): exact ids and scores."""
import ctypes as C

import numpy as np
import pytest

import util
from oracle import oracle as orc
from oramacore_amd import _native as N

pytestmark = pytest.mark.gpu


def gpu_top_n(ctx, doc, score, k):
    lib = N.load()
    doc = np.ascontiguousarray(doc, dtype=np.uint64)
    score = np.ascontiguousarray(score, dtype=np.float32)
    out_ids = np.zeros(max(k, 1), dtype=np.uint64)
    out_sc = np.zeros(max(k, 1), dtype=np.float32)
    n = C.c_uint32()
    N.check(lib.orama_top_n(ctx.handle, doc.ctypes.data, score.ctypes.data, doc.shape[0], k,
                            out_ids.ctypes.data, out_sc.ctypes.data, C.byref(n)))
    return out_ids[:n.value], out_sc[:n.value]


@pytest.mark.parametrize("n", [1, 2, 100, 4096, 4097, 8192, 8193, 50_000, 131_072, 131_073, 1_000_003])
@pytest.mark.parametrize("k", [1, 10, 100, 257, 1000, 4096])
def test_top_n_matches_oracle_bit_exact(ctx, n, k):
    rng = np.random.default_rng(n * 31 + k)
    doc = rng.permutation(n * 3)[:n].astype(np.uint64)
    score = rng.standard_normal(n).astype(np.float32)
    # heavy ties + NaN + signed zeros + infinities
    score[rng.random(n) < 0.3] = np.float32(1.5)
    score[rng.random(n) < 0.05] = np.nan
    score[rng.random(n) < 0.02] = np.float32(-0.0)
    score[rng.random(n) < 0.02] = np.float32(0.0)
    if n > 10:
        score[3] = np.inf
        score[7] = -np.inf
    g_ids, g_sc = gpu_top_n(ctx, doc, score, k)
    o_ids, o_sc = orc.top_n(doc, score, k)
    assert np.array_equal(g_ids, o_ids)
    assert np.array_equal(g_sc.view(np.uint32) | np.uint32(0), o_sc.view(np.uint32)) or np.array_equal(g_sc, o_sc)


def test_top_n_all_equal_and_all_nan(ctx):
    n = 300_000
    doc = np.arange(n, dtype=np.uint64)[::-1].copy()
    g_ids, g_sc = gpu_top_n(ctx, doc, np.full(n, 2.0, dtype=np.float32), 50)
    assert g_ids.tolist() == list(range(50)) and np.all(g_sc == 2.0)
    g_ids, g_sc = gpu_top_n(ctx, doc, np.full(n, np.nan, dtype=np.float32), 50)
    assert len(g_ids) == 0
    g_ids, g_sc = gpu_top_n(ctx, doc[:10], np.arange(10, dtype=np.float32), 50)
    assert g_ids.tolist() == doc[:10][::-1].tolist()
