"""Micro-batcher (orama_batcher_*): concurrent single-query callers coalesced into shared corpus passes.
Every caller must get exactly what a lone orama_vec_search(q=1) returns (same ids, same distances)."""
import threading

import numpy as np
import pytest

import oramacore_amd as oa
import util
from oracle import oracle as orc
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


def test_filtered_requests_share_passes_and_match_the_oracle(ctx):
    """While deletes are pending every search carries the NOT-deleted predicate (index/filter.rs:344-392 ->
    search_with_filter, embedding_field.rs:255-262).  Requests with the SAME resident bitmap share corpus passes;
    requests with another filter form their own batches; every answer is checked against the ORACLE's filtered scan
    (on the fp16-rounded vectors, 1e-4), not only against a solo run."""
    n, d, k = 20000, 384, 15
    corpus = util.gaussian_rows(n, d, seed=41)
    doc_ids = np.arange(n, dtype=np.uint64)
    st = oa.EmbeddingFieldStorage(ctx, dimensions=d, dtype=N.DTYPE_F16)
    st.insert_rows(doc_ids, corpus)
    c16 = corpus.astype(np.float16).astype(np.float32)
    deleted = np.arange(0, n, 7)
    not_deleted = oa.AllowBitmap.from_mask(~np.isin(np.arange(n), deleted))
    category = oa.AllowBitmap.from_mask(np.arange(n) % 5 == 2)
    res_nd, res_cat = not_deleted.to_device(ctx), category.to_device(ctx)
    n_clients = 96
    queries = util.gaussian_rows(n_clients, d, seed=42)
    filt = [res_cat if i % 8 == 0 else (None if i % 8 == 1 else res_nd) for i in range(n_clients)]
    b = oa.SearchBatcher(st, max_batch=64, max_wait_us=30000)
    got, errs = [None] * n_clients, []

    def client(i):
        try:
            got[i] = b.search(queries[i], k, allow=filt[i])
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    threads = [threading.Thread(target=client, args=(i,)) for i in range(n_clients)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errs, errs
    for i in range(n_clients):
        host = None if filt[i] is None else (category if filt[i] is res_cat else not_deleted)
        q16 = queries[i].astype(np.float16).astype(np.float32)
        full = orc.distances(c16, q16).astype(np.float64)
        if host is not None:
            mask = np.array([host.contains(int(x)) for x in range(n)])
            full[~mask] = np.nan
        ids, dist = got[i]
        util.assert_topk_sound(ids, dist, full, k, 1e-4, f"client {i}")
        if host is not None:
            assert all(host.contains(int(x)) for x in ids.tolist())
    s = b.stats()
    assert s["requests"] == n_clients and s["batches"] < n_clients // 2 and s["largest_batch"] > 8
    b.close()
    res_nd.close()
    res_cat.close()
    st.close()
