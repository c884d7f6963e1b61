"""Micro-batcher (orama_batcher_*): concurrent single-query callers coalesced into shared corpus passes.
Every caller must get exactly what a lone orama_vec_search(q=1) returns (same ids, same distances)."""
import threading

import numpy as np
import pytest

import oramacore_amd as oa
import util
from oramacore_amd import _native as N

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [N.DTYPE_F16, N.DTYPE_F32], ids=["f16", "f32"])
def test_batched_answers_equal_solo_answers(ctx, dtype):
    n, d, n_clients = 20000, 128, 48
    corpus = util.gaussian_rows(n, d, seed=3)
    st = oa.EmbeddingFieldStorage(ctx, dimensions=d, dtype=dtype)
    st.insert_rows(np.arange(n, dtype=np.uint64) * 3 + 1, corpus)
    queries = util.gaussian_rows(n_clients, d, seed=4)
    ks = [1 + (7 * i) % 60 for i in range(n_clients)]
    solo = []
    for i in range(n_clients):
        ids, dist, cnt = st.storage_search(queries[i], ks[i])
        solo.append((ids[0, :cnt[0]].copy(), dist[0, :cnt[0]].copy()))
    b = oa.SearchBatcher(st, max_batch=16, max_wait_us=2000)
    got = [None] * n_clients
    errs = []

    def client(i):
        try:
            for _ in range(3):
                got[i] = b.search(queries[i], ks[i])
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    threads = [threading.Thread(target=client, args=(i,)) for i in range(n_clients)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errs, errs
    for i in range(n_clients):
        assert got[i][0].tolist() == solo[i][0].tolist(), i
        assert np.array_equal(got[i][1], solo[i][1]), i
    s = b.stats()
    assert s["requests"] == 3 * n_clients
    assert s["batches"] < s["requests"] and 1 < s["largest_batch"] <= 16
    b.close()
    st.close()


def test_batcher_limits_and_errors(ctx):
    st = oa.EmbeddingFieldStorage(ctx, dimensions=8)
    st.insert_rows(np.arange(5, dtype=np.uint64), util.gaussian_rows(5, 8, seed=1))
    b = oa.SearchBatcher(st)
    ids, dist = b.search(np.ones(8, np.float32), 0)
    assert len(ids) == 0
    ids, dist = b.search(np.ones(8, np.float32), 100)  # fewer rows than k
    assert len(ids) == 5 and np.all(np.diff(dist) >= 0)
    with pytest.raises(oa.OramaError):
        b.search(np.ones(8, np.float32), 5000)  # above the supported limit
    with pytest.raises(oa.OramaError):
        oa.SearchBatcher(st, max_batch=0)
    b.close()
    st.close()


def test_wide_batches_through_the_batcher(ctx):
    """max_batch > 64 on an fp16 store: batches of 65..256 requests take the GEMM-tiled pass (K2c) and every caller
    still gets its solo answer."""
    n, d, n_clients = 30000, 384, 160
    st = oa.EmbeddingFieldStorage(ctx, dimensions=d, dtype=N.DTYPE_F16)
    st.insert_rows(np.arange(n, dtype=np.uint64), util.gaussian_rows(n, d, seed=13))
    queries = util.gaussian_rows(n_clients, d, seed=14)
    solo = []
    for i in range(n_clients):
        ids, dist, cnt = st.storage_search(queries[i], 20)
        solo.append((ids[0, :cnt[0]].copy(), dist[0, :cnt[0]].copy()))
    b = oa.SearchBatcher(st, max_batch=256, max_wait_us=20000)
    got = [None] * n_clients
    errs = []

    def client(i):
        try:
            got[i] = b.search(queries[i], 20)
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    threads = [threading.Thread(target=client, args=(i,)) for i in range(n_clients)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errs, errs
    for i in range(n_clients):
        assert got[i][0].tolist() == solo[i][0].tolist(), i
        assert np.array_equal(got[i][1], solo[i][1]), i
    assert b.stats()["largest_batch"] > 64
    b.close()
    st.close()
