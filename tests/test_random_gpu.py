"""GPU: the HIP path against the oracle on RANDOM inputs (hypothesis), through the C ABI — the same strategies as
tests/test_oracle_random.py: special floats (zero, denormal, inf, NaN, negative), empty maps, up to 40 tokens
(the u32 mask wraps at 32), thresholds, OMC maps; bit-exact ids, scores and counts."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

import oramacore_amd as oa
from oracle import oracle as orc
from oramacore_amd import fulltext as ft
from test_oracle_random import contributions, two_maps

pytestmark = pytest.mark.gpu
CTX = None


def context():
    global CTX
    if CTX is None:
        CTX = oa.Context(0)
    return CTX


def bits(x):
    """Bit patterns, with -0.0 folded onto +0.0: K4's order-preserving key canonicalises the sign of zero so that
    +-0 tie exactly like `NotNan<f32>` compares them, and the value is rebuilt from the key — a score of -0.0 comes
    back as +0.0 (declared deviation; the scoring pipeline itself never produces -0.0)."""
    a = np.asarray(x, dtype=np.float32).copy()
    a[a == 0] = 0.0
    return a.view(np.uint32).tolist()


@settings(max_examples=120, deadline=None, suppress_health_check=[HealthCheck.too_slow])
@given(contributions(), st.integers(1, 50), st.dictionaries(st.integers(0, 60), st.sampled_from([0.25, 0.5, 2.0, 10.0]),
                                                           max_size=4))
def test_bm25_score_random(c, top_k, omc):
    entries, n_tokens, n_docs, thr = c
    if not entries:
        entries = [(0, np.zeros(0, np.uint64), np.zeros(0, np.float32))]
    ids, sc, count = ft.bm25_score(context(), entries, n_tokens, float(n_docs), top_k, thr, omc=omc or None)
    od, os_ = orc.search_full_text(entries, n_tokens, float(n_docs), 1.2, thr)
    if omc:
        os_ = orc.apply_omc(od, os_, sorted(omc), [omc[d] for d in sorted(omc)])
    td, ts = orc.top_n(od, os_, top_k)
    assert count == len(od)
    assert ids.tolist() == td.tolist()
    assert bits(sc) == bits(ts)


@settings(max_examples=120, deadline=None, suppress_health_check=[HealthCheck.too_slow])
@given(two_maps())
def test_hybrid_combine_and_top_n_random(m):
    vec, ftm, n = m
    n = max(n, 1)
    ids, sc, count = ft.hybrid_combine(context(), vec, ftm, n)
    od, os_ = orc.normalize_and_combine(list(vec), list(vec.values()), list(ftm), list(ftm.values()))
    td, ts = orc.top_n(od, os_, n)
    assert count == len(od)
    assert ids.tolist() == td.tolist()
    assert bits(sc) == bits(ts)
    if ftm:
        i2, s2 = ft.top_n(context(), ftm, n)
        d2, v2 = orc.top_n(sorted(ftm), [ftm[d] for d in sorted(ftm)], n)
        assert i2.tolist() == d2.tolist() and bits(s2) == bits(v2)


@st.composite
def small_index(draw):
    n_docs = draw(st.integers(1, 120))
    id_mul = draw(st.sampled_from([1, 1, 3]))
    doc_ids = np.arange(n_docs, dtype=np.uint64) * np.uint64(id_mul) + np.uint64(draw(st.integers(0, 5)))
    n_fields = draw(st.integers(1, 3))
    lens = [np.array(draw(st.lists(st.integers(1, 300), min_size=n_docs, max_size=n_docs)), dtype=np.uint32)
            for _ in range(n_fields)]
    n_lists = draw(st.integers(1, 8))
    lists = []
    for _ in range(n_lists):
        f = draw(st.integers(0, n_fields - 1))
        pos = sorted(draw(st.sets(st.integers(0, n_docs - 1), max_size=min(n_docs, 40))))
        tf = [draw(st.integers(1, 9)) for _ in pos]
        lists.append((f, np.array(pos, dtype=np.int64), np.array(tf, dtype=np.uint32)))
    n_tokens = draw(st.integers(1, 6))
    refs = [(draw(st.integers(0, n_tokens - 1)), draw(st.integers(0, n_lists - 1)),
             draw(st.sampled_from([1.0, 1.0, 2.0, 0.5]))) for _ in range(draw(st.integers(0, 10)))]
    refs.sort(key=lambda r: r[0])  # entries of a token are consumed in the order given
    thr = draw(st.one_of(st.none(), st.integers(0, 4)))
    allow = draw(st.one_of(st.none(), st.lists(st.booleans(), min_size=n_docs, max_size=n_docs)))
    omc = draw(st.dictionaries(st.integers(0, n_docs - 1), st.sampled_from([0.25, 2.0, 10.0]), max_size=3))
    vec = draw(st.one_of(st.none(), st.dictionaries(st.integers(0, n_docs - 1),
                                                    st.floats(min_value=-0.5, max_value=1.0, width=32), max_size=5)))
    return doc_ids, lens, lists, n_tokens, refs, thr, allow, omc, vec


@settings(max_examples=80, deadline=None, suppress_health_check=[HealthCheck.too_slow, HealthCheck.data_too_large])
@given(small_index(), st.integers(1, 40))
def test_resident_postings_random(ix, top_k):
    doc_ids, lens, lists, n_tokens, refs, thr, allow, omc, vec = ix
    n_docs = len(doc_ids)
    avg = [float(np.float32(l.astype(np.float64).mean())) for l in lens]
    store = ft.PostingsStore(context())
    store.build(doc_ids, avg, [ft.PostingList(field=f, docs=doc_ids[pos], tf=tf, field_len=lens[f][pos])
                               for f, pos, tf in lists])
    omc_ids = {int(doc_ids[d]): m for d, m in omc.items()}
    if omc_ids:
        store.set_omc(omc_ids)
    bm = None
    if allow is not None:
        bm = oa.AllowBitmap(int(doc_ids.max()) + 1, doc_ids[np.array(allow, dtype=bool)])
    vmap = None if vec is None else {int(doc_ids[d]): float(s) for d, s in vec.items()}
    ids, sc, count = store.search(refs, n_tokens, float(n_docs), top_k, thr, allow=bm, apply_omc=bool(omc_ids),
                                  vector=vmap)
    # oracle: contributions with host-side ntf, filter applied to postings
    entries = []
    for tok, l, boost in refs:
        f, pos, tf = lists[l]
        keep = np.ones(len(pos), dtype=bool) if allow is None else np.array(allow, dtype=bool)[pos]
        ntf = np.array([np.float32(boost) * orc.bm25f_normalized_tf(int(t), int(lens[f][p]), avg[f], 0.75)
                        for p, t in zip(pos[keep], tf[keep])], dtype=np.float32)
        entries.append((tok, doc_ids[pos[keep]], ntf))
    od, os_ = orc.search_full_text(entries, n_tokens, float(n_docs), 1.2, thr)
    if vmap is not None:
        od, os_ = orc.normalize_and_combine(sorted(vmap), [vmap[d] for d in sorted(vmap)], od, os_)
    if omc_ids:
        os_ = orc.apply_omc(od, os_, sorted(omc_ids), [omc_ids[d] for d in sorted(omc_ids)])
    td, ts = orc.top_n(od, os_, top_k)
    assert count == len(od)
    assert ids.tolist() == td.tolist()
    assert bits(sc) == bits(ts)
    if vmap is None:  # the plain search ran on the range-partitioned scorer (K3r): K3 must give the same answer
        context().set_bm25_ranges(False)
        try:
            ids3, sc3, count3 = store.search(refs, n_tokens, float(n_docs), top_k, thr, allow=bm, apply_omc=bool(omc_ids))
        finally:
            context().set_bm25_ranges(True)
        assert count3 == count and ids3.tolist() == ids.tolist() and bits(sc3) == bits(sc)
    store.close()


@settings(max_examples=40, deadline=None, suppress_health_check=[HealthCheck.too_slow, HealthCheck.data_too_large])
@given(st.integers(1, 2500), st.sampled_from([3, 8, 64, 100, 129, 384, 768, 1024, 1100]), st.integers(1, 11),
       st.integers(1, 60), st.booleans(), st.booleans(), st.booleans(), st.integers(0, 2**31))
def test_vector_scan_random(n, d, nq, k, f16, use_filter, use_deletes, seed):
    import util

    rng = np.random.default_rng(seed)
    corpus = util.gaussian_rows(n, d, seed=seed % 100003)
    doc_ids = np.arange(n, dtype=np.uint64) * 2 + 3
    st_ = oa.EmbeddingFieldStorage(context(), dimensions=d, dtype=oa.DTYPE_F16 if f16 else oa.DTYPE_F32)
    st_.insert_rows(doc_ids, corpus)
    dead = np.zeros(n, dtype=bool)
    if use_deletes and n > 2:
        for r in rng.choice(n, size=min(3, n - 1), replace=False):
            st_.delete(int(doc_ids[r]))
            dead[r] = True
    allow_mask = rng.random(n) < 0.7 if use_filter else np.ones(n, dtype=bool)
    bm = oa.AllowBitmap(int(doc_ids.max()) + 1, doc_ids[allow_mask]) if use_filter else None
    queries = util.gaussian_rows(nq, d, seed=(seed + 1) % 100003)
    ids, dist, cnt = st_.storage_search(queries, k, bm)
    ref_rows = corpus.astype(np.float16).astype(np.float32) if f16 else corpus
    live = allow_mask & ~dead
    for qi in range(nq):
        qv = queries[qi].astype(np.float16).astype(np.float32) if f16 else queries[qi]
        full = orc.distances(ref_rows, qv).astype(np.float64)
        full[~live] = np.nan
        m = int(cnt[qi])
        assert m == min(k, int(live.sum()))
        util.assert_topk_sound((ids[qi, :m] - 3) // 2, dist[qi, :m], full, k, 1e-4, f"n={n} d={d} q{qi} f16={f16}")
    st_.close()


@settings(max_examples=40, deadline=None, suppress_health_check=[HealthCheck.too_slow, HealthCheck.data_too_large])
@given(st.integers(1, 3000), st.sampled_from([4, 64, 100, 128, 384, 768, 1000, 1024]), st.integers(1, 12), st.integers(1, 300),
       st.booleans(), st.booleans(), st.integers(0, 2**31))
def test_two_stage_equals_plain_scan_random(n, d, nq, k, use_filter, use_deletes, seed):
    """fp32 rows + fp16 shadow against a plain fp32 store fed the same rows: ids, distance bits and counts."""
    import util

    ctx = context()
    ctx.set_two_stage(True, always=True)
    try:
        rng = np.random.default_rng(seed)
        corpus = util.gaussian_rows(n, d, seed=seed % 100003) * rng.uniform(0.2, 3.0, size=(n, 1)).astype(np.float32)
        if n > 10:
            corpus[n // 2] = corpus[n // 3]  # an exact duplicate: tie resolved by (doc, row) on both sides
        doc_ids = np.arange(n, dtype=np.uint64) * 2 + 3
        plain = oa.EmbeddingFieldStorage(ctx, dimensions=d, dtype=oa.DTYPE_F32)
        shadow = oa.EmbeddingFieldStorage(ctx, dimensions=d, dtype=oa.DTYPE_F32_SHADOW16)
        for st_ in (plain, shadow):
            st_.insert_rows(doc_ids, corpus)
        if use_deletes and n > 2:
            for r in rng.choice(n, size=min(3, n - 1), replace=False):
                plain.delete(int(doc_ids[r]))
                shadow.delete(int(doc_ids[r]))
        bm = oa.AllowBitmap(int(doc_ids.max()) + 1, doc_ids[rng.random(n) < 0.7]) if use_filter else None
        queries = util.gaussian_rows(nq, d, seed=(seed + 1) % 100003)
        a = plain.storage_search(queries, k, bm)
        b = shadow.storage_search(queries, k, bm)
        assert a[2].tolist() == b[2].tolist()
        for j in range(nq):
            m = int(a[2][j])
            assert a[0][j, :m].tolist() == b[0][j, :m].tolist(), (n, d, j)
            assert a[1][j, :m].view(np.uint32).tolist() == b[1][j, :m].view(np.uint32).tolist(), (n, d, j)
        assert shadow.info()["two_stage_queries"] == nq
        plain.close()
        shadow.close()
    finally:
        ctx.set_two_stage(True)
