"""The envelope limits listed at the top of include/orama_hip.h: a request beyond one of them returns
ORAMA_ERR_UNSUPPORTED (4) — never a truncated answer — and a malformed one ORAMA_ERR_INVALID (1)."""
import numpy as np
import pytest

import oramacore_amd as oa
from oramacore_amd import _native as N
from oramacore_amd import fulltext as ft

pytestmark = pytest.mark.gpu

UNSUPPORTED, INVALID = 4, 1


@pytest.fixture(scope="module")
def ctx():
    c = oa.Context(0)
    yield c
    c.close()


def status_of(fn):
    with pytest.raises(N.OramaError) as e:
        fn()
    return e.value.status


def test_vector_limit(ctx):
    rng = np.random.default_rng(0)
    for dtype in (N.DTYPE_F32, N.DTYPE_F16):
        st = oa.EmbeddingFieldStorage(ctx, dimensions=16, dtype=dtype)
        st.insert_rows(np.arange(5000, dtype=np.uint64), rng.standard_normal((5000, 16)).astype(np.float32))
        q = rng.standard_normal((3, 16)).astype(np.float32)
        ids, dist, cnt = st.storage_search(q, 4096)  # the largest limit served
        assert cnt.tolist() == [4096] * 3 and np.all(np.diff(dist, axis=1) >= 0)
        assert status_of(lambda: st.storage_search(q, 4097)) == UNSUPPORTED
        st.close()
    assert status_of(lambda: oa.EmbeddingFieldStorage(ctx, dimensions=65537)) == UNSUPPORTED
    assert status_of(lambda: oa.EmbeddingFieldStorage(ctx, dimensions=0)) == INVALID
    assert status_of(lambda: oa.EmbeddingFieldStorage(ctx, dimensions=8, metric=7)) == INVALID


def test_fulltext_limits(ctx):
    docs = np.arange(10, dtype=np.uint64)
    ntf = np.ones(10, dtype=np.float32)
    ids, sc, count = ft.bm25_score(ctx, [(63, docs, ntf)], 64, 100.0, 4096)  # both at their maxima
    assert count == 10 and len(ids) == 10
    assert status_of(lambda: ft.bm25_score(ctx, [(0, docs, ntf)], 65, 100.0, 10)) == UNSUPPORTED
    assert status_of(lambda: ft.bm25_score(ctx, [(0, docs, ntf)], 1, 100.0, 4097)) == UNSUPPORTED
    assert status_of(lambda: ft.bm25_score(ctx, [(0, docs, ntf)], 0, 100.0, 10)) == INVALID
    assert status_of(lambda: ft.bm25_score(ctx, [(5, docs, ntf)], 2, 100.0, 10)) == INVALID  # token index >= n_tokens
    # the whole map has no size limit
    n = 50_000
    big = np.arange(n, dtype=np.uint64)
    full = ft.bm25_score_map(ctx, [(0, big, np.ones(n, dtype=np.float32))], 1, float(n))
    assert len(full) == n
    assert status_of(lambda: ft.top_n(ctx, (big, np.ones(n, np.float32)), 4097)) == UNSUPPORTED


def test_group_and_range_limits(ctx):
    docs = np.arange(100, dtype=np.uint64)
    store = ft.PostingsStore(ctx)
    store.build(docs, [4.0], [ft.PostingList(field=0, docs=docs, tf=np.ones(100), field_len=np.full(100, 4))])
    smap = store.search_scores([(0, 0, 1.0)], 1, 100.0, 10)
    assert smap.hits[2] == 100
    field = ft.FacetField.buckets(store, [docs[:50], docs[50:]])
    assert status_of(lambda: smap.group_top(field, 1025)) == UNSUPPORTED
    assert status_of(lambda: smap.group_top(field, 0)) == INVALID
    nums = ft.FacetField.numbers(store, docs, np.arange(100, dtype=np.float64))
    # any number of ranges (the library counts 64 per launch)
    assert smap.facet_count_ranges(nums, [(float(i), float(i + 1)) for i in range(99)]).tolist() == [2] * 99
    field.close()
    nums.close()
    smap.close()
    store.close()
