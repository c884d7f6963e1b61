"""GPU parity tests of the vector path (K1 scan + K4 top-k) through the C ABI.

Bar (BASELINE.json north_star): cosine scores within 1e-4 of the reference's fp32 arithmetic; ids
identical wherever scores are separated by more than the fp32 summation-order noise.
"""
import numpy as np
import pytest

import oramacore_amd as oa
import util
from oracle import oracle as orc
from test_oracle_golden import multirow_inputs

pytestmark = pytest.mark.gpu

TOL = 1e-4  # north_star: "cosine scores within 1e-4 fp32"


def make_store(ctx, corpus, row_doc=None, metric=oa.METRIC_COSINE):
    n, d = corpus.shape
    st = oa.EmbeddingFieldStorage(ctx, dimensions=d, metric=metric)
    ids = np.arange(n, dtype=np.uint64) if row_doc is None else row_doc
    assert st.insert_rows(ids, corpus) == n
    return st


@pytest.mark.parametrize("d", [384, 768])
def test_cosine_small_vs_golden_and_oracle(ctx, d):
    g = np.load(util.GOLDEN / "cosine_small.npz")
    n, nq, k = 4096, 8, 100
    corpus = util.det_matrix(n, d, seed=1000 + d)
    queries = util.det_matrix(nq, d, seed=2000 + d)
    st = make_store(ctx, corpus)
    ids, dist, cnt = st.storage_search(queries, k)  # batch entry: 8 independent queries
    assert cnt.tolist() == [k] * nq
    for qi in range(nq):
        util.assert_ranked_equal(ids[qi], dist[qi], g[f"ids_{d}"][qi], g[f"dist32_{d}"][qi], TOL, f"q{qi}")
        assert np.max(np.abs(dist[qi].astype(np.float64) - g[f"dist64_{d}"][qi])) < TOL
        full = orc.distances(corpus, queries[qi])
        util.assert_topk_sound(ids[qi], dist[qi], full, k, TOL, f"q{qi}")
    # single-query entry (the reference's shape) gives the same answer as the batch entry
    ids1, dist1, cnt1 = st.storage_search(queries[3], k)
    assert np.array_equal(ids1[0], ids[3]) and np.array_equal(dist1[0], dist[3])
    st.close()


@pytest.mark.parametrize("d,n", [(1024, 3000), (384, 1), (768, 63), (100, 777), (7, 50), (2048, 130), (96, 5000)])
def test_dims_and_ragged_sizes(ctx, d, n):
    """Every reference model dimension (384/768/1024, src/python/embeddings.rs:52-63) plus odd dims that
    take the generic kernel, and row counts that do not fill the last wave group."""
    corpus = util.gaussian_rows(n, d, seed=d * 7 + n)
    q = util.gaussian_rows(1, d, seed=d + 1)[0]
    st = make_store(ctx, corpus)
    for k in (1, 10, 100):
        ids, dist, cnt = st.storage_search(q, k)
        m = int(cnt[0])
        assert m == min(k, n)
        full = orc.distances(corpus, q)
        util.assert_topk_sound(ids[0, :m], dist[0, :m], full, k, TOL, f"d={d} n={n} k={k}")
    st.close()


def test_empty_store_and_limit_zero(ctx):
    st = oa.EmbeddingFieldStorage(ctx, oa.Model.BGESmall)
    q = np.ones(384, dtype=np.float32)
    ids, dist, cnt = st.storage_search(q, 10)
    assert cnt[0] == 0
    st.insert(5, q)
    ids, dist, cnt = st.storage_search(q, 0)
    assert cnt[0] == 0
    ids, dist, cnt = st.storage_search(q, 10)
    assert cnt[0] == 1 and ids[0, 0] == 5 and abs(dist[0, 0]) < 1e-6
    st.close()


def test_insert_rejects_invalid_rows(ctx):
    """EmbeddingIndexer::index_vec_vec -> None (embedding_field.rs:232-237): zero / non-finite rows."""
    d = 384
    rows = util.gaussian_rows(6, d, seed=3)
    rows[1] = 0.0
    rows[3, 17] = np.nan
    rows[4, 0] = np.inf
    st = oa.EmbeddingFieldStorage(ctx, dimensions=d)
    assert st.insert_rows(np.arange(6, dtype=np.uint64), rows) == 3
    assert st.info()["num_embeddings"] == 3
    ids, dist, cnt = st.storage_search(rows[0], 10)
    assert sorted(ids[0, :cnt[0]].tolist()) == [0, 2, 5]
    for j in range(6):
        assert orc.row_is_valid(rows[j]) == (j in (0, 2, 5))
    st.close()


def test_tie_rule_matches_oracle_exactly(ctx):
    """Duplicated rows under shuffled doc ids: equal distances are ordered by doc id, and the cut at k
    keeps the lowest rows (declared rule; reference unpinned — SURVEY F6)."""
    g = util.load_json("cosine_ties.json")
    d, base_n = 384, 64
    base = util.det_matrix(base_n, d, seed=77)
    order = np.argsort(util.hash_u64(np.arange(base_n * 4, dtype=np.uint64) + np.uint64(5)), kind="stable")
    corpus = np.concatenate([base] * 4, axis=0)[order]
    row_doc = (np.arange(base_n * 4, dtype=np.uint64) * np.uint64(3) + np.uint64(7))[
        np.argsort(util.hash_u64(np.arange(base_n * 4, dtype=np.uint64) + np.uint64(11)), kind="stable")]
    q = util.det_matrix(1, d, seed=78)[0]
    st = make_store(ctx, corpus, row_doc)
    for k, exp in g.items():
        ids, dist, cnt = st.storage_search(q, int(k))
        m = int(cnt[0])
        assert m == len(exp["ids"])
        exp_dist = np.array(exp["dist_bits"], dtype=np.uint32).view(np.float32)
        assert np.max(np.abs(dist[0, :m] - exp_dist)) <= TOL
        # duplicates are bit-identical rows => bit-identical device distances => exact tie groups:
        # within each group of 4 the ids must come out ascending, like the oracle's.
        got = ids[0, :m].tolist()
        dd = dist[0, :m]
        i = 0
        while i < m:
            j = i
            while j < m and dd[j] == dd[i]:
                j += 1
            assert got[i:j] == sorted(got[i:j])
            i = j
        util.assert_ranked_equal(ids[0, :m], dist[0, :m], exp["ids"], exp_dist, TOL, f"ties k={k}")
    st.close()


def test_multirow_delete_filter_epilogue(ctx):
    """N rows per doc, tombstones, allow-bitmap, then the in-tree epilogue (embedding_field.rs:268-276)
    with and without the E5 rescale (python/embeddings.rs:71-92)."""
    g = util.load_json("cosine_multirow.json")
    corpus, row_doc, q, dead, allow, dead_docs = multirow_inputs()
    bm = oa.AllowBitmap.from_mask(allow)
    for name, use_dead, use_filter in (("plain", 0, 0), ("dead", 1, 0), ("filter", 0, 1), ("dead_filter", 1, 1)):
        st = make_store(ctx, corpus, row_doc)
        if use_dead:
            for dd in dead_docs:
                st.delete(dd)
            assert st.has_pending_ops()
        for k in (5, 10, 50):
            exp = g[f"{name}_k{k}"]
            ids, dist, cnt = st.storage_search(q, k, bm if use_filter else None)
            m = int(cnt[0])
            util.assert_ranked_equal(ids[0, :m], dist[0, :m], exp["ids"], exp["dist"], TOL, f"{name} k={k}")
            for model, is_e5 in ((oa.Model.BGESmall, 0), (oa.Model.MultilingualE5Small, 1)):
                for ms in (0.0, 0.7):
                    st._model = model
                    out = {}
                    st.search(oa.VectorSearchParams(target=q, similarity=ms, limit=k,
                                                    filtered_doc_ids=bm if use_filter else None), out)
                    expm = exp[f"map_e5{is_e5}_min{ms}"]
                    # scores near the cut-off may flip with 1e-4 noise: compare docs clearly inside/outside
                    for dk, v in expm.items():
                        if v > ms + 5 * TOL or ms == 0.0:
                            assert int(dk) in out and abs(float(out[int(dk)]) - v) <= 20 * TOL, (name, k, dk)
                    for dk, v in out.items():
                        assert str(dk) in expm or float(v) <= ms + 5 * TOL
        if use_dead:
            # compaction drops the tombstones and must not change results
            before = st.storage_search(q, 50, bm if use_filter else None)
            st.compact(7)
            assert not st.has_pending_ops() and st.current_version_number() == 7
            after = st.storage_search(q, 50, bm if use_filter else None)
            assert np.array_equal(before[0], after[0]) and np.array_equal(before[2], after[2])
            assert np.allclose(before[1], after[1], atol=1e-6)
        st.close()


def test_l2_extension(ctx):
    """Squared-L2 metric (build-side extension) against the oracle's direct form."""
    n, d = 5000, 384
    corpus = util.gaussian_rows(n, d, seed=11)
    q = util.gaussian_rows(1, d, seed=12)[0]
    st = make_store(ctx, corpus, metric=oa.METRIC_L2SQ)
    ids, dist, cnt = st.storage_search(q, 100)
    full = orc.distances(corpus, q, metric=1)
    util.assert_topk_sound(ids[0], dist[0], full, 100, 1e-4, "l2")
    st.close()


def test_zero_query_declared_rule(ctx):
    corpus = util.gaussian_rows(100, 384, seed=5)
    st = make_store(ctx, corpus)
    ids, dist, cnt = st.storage_search(np.zeros(384, dtype=np.float32), 5)
    assert cnt[0] == 5 and np.all(dist[0] == 1.0) and ids[0].tolist() == [0, 1, 2, 3, 4]
    st.close()


def test_search_is_idempotent_and_incremental(ctx):
    """Appending rows keeps earlier rows' distances; repeated searches are bit-identical."""
    d = 768
    a = util.gaussian_rows(3000, d, seed=21)
    b = util.gaussian_rows(2000, d, seed=22)
    q = util.gaussian_rows(1, d, seed=23)[0]
    st = make_store(ctx, a)
    r1 = st.storage_search(q, 50)
    r2 = st.storage_search(q, 50)
    assert all(np.array_equal(x, y) for x, y in zip(r1, r2))
    st.insert_rows(np.arange(3000, 5000, dtype=np.uint64), b)
    ids, dist, cnt = st.storage_search(q, 50)
    full = orc.distances(np.concatenate([a, b]), q)
    util.assert_topk_sound(ids[0], dist[0], full, 50, TOL, "after append")
    st.close()


def test_synthetic_fill_properties_1m(ctx):
    """BASELINE configs[1] shape (1 M x 384 fp32, k = 100) through size-independent properties:
    sortedness, planted exact matches on top, top-k soundness on the rows read back, and
    merge-of-halves == whole (the shard/merge identity used by the multi-GPU path)."""
    n, d, k = 1_000_000, 384, 100
    st = oa.EmbeddingFieldStorage(ctx, oa.Model.BGESmall, reserve_rows=n + 16)
    st.fill_synthetic(n, seed=0xC0FFEE, first_doc_id=0)
    q = util.gaussian_rows(1, d, seed=0xBEEF)[0]
    # plant 3 scaled copies of the query at the end: cosine distance ~ 0
    plant = np.stack([q * np.float32(s) for s in (0.5, 1.0, 3.0)])
    st.insert_rows(np.arange(n, n + 3, dtype=np.uint64), plant)
    ids, dist, cnt = st.storage_search(q, k)
    assert cnt[0] == k
    assert sorted(ids[0, :3].tolist()) == [n, n + 1, n + 2] and np.all(np.abs(dist[0, :3]) < 1e-6)
    assert np.all(np.diff(dist[0]) >= 0)
    # reported distances agree with the oracle on the rows themselves (doc id == row here)
    rows, docs = st.get_rows(ids[0])
    assert np.array_equal(docs, ids[0])
    od = np.array([orc.distances(rows[i:i + 1], q)[0] for i in range(k)])
    assert np.max(np.abs(od - dist[0])) <= TOL
    # rows have the declared norm distribution U(0.5, 2)
    norms = np.linalg.norm(rows[3:], axis=1)
    assert norms.min() >= 0.49 and norms.max() <= 2.01
    # a random sample of other rows must not beat the k-th distance
    rng = np.random.default_rng(1)
    sample = rng.choice(n, size=20000, replace=False).astype(np.uint64)
    srows, _ = st.get_rows(sample)
    sd = orc.distances(srows, q)
    inside = set(ids[0].tolist())
    for r, v in zip(sample.tolist(), sd.tolist()):
        assert r in inside or v >= dist[0, -1] - 2 * TOL
    # shard/merge identity: top-k(all) == merge(top-k(even docs), top-k(odd docs))
    even = oa.AllowBitmap.from_mask(np.arange(n + 3) % 2 == 0)
    odd = oa.AllowBitmap.from_mask(np.arange(n + 3) % 2 == 1)
    ie, de, ce = st.storage_search(q, k, even)
    io, do, co = st.storage_search(q, k, odd)
    assert np.all(ie[0] % 2 == 0) and np.all(io[0] % 2 == 1)
    md, ms = orc.top_n(np.concatenate([ie[0], io[0]]), -np.concatenate([de[0], do[0]]), k)
    assert np.array_equal(md, ids[0]) and np.array_equal(-ms, dist[0])
    st.close()


def test_concurrent_searches_with_inserts_and_deletes(ctx):
    """Re-entrancy contract of the ABI (the reference calls search(&self) from many tokio workers while
    update_data(&self) inserts — collection.rs:846-884, index/mod.rs:1436): 8 threads search while another
    thread appends rows and tombstones documents that are far from the queries; every search must return the
    same exact answer."""
    import threading

    d, n = 384, 60_000
    corpus = util.gaussian_rows(n, d, seed=101)
    queries = util.gaussian_rows(8, d, seed=102)
    for qi in range(8):  # plant clear winners so that appended noise never enters the top-10
        for j in range(10):
            corpus[qi * 100 + j] = queries[qi] * np.float32(1 + j) + util.gaussian_rows(1, d, seed=qi * 31 + j)[0] * np.float32(0.05 * (j + 1))
    st = make_store(ctx, corpus)
    expected = [st.storage_search(queries[qi], 10) for qi in range(8)]
    errors = []
    stop = threading.Event()

    def searcher(qi):
        try:
            for _ in range(25):
                ids, dist, cnt = st.storage_search(queries[qi], 10)
                assert np.array_equal(ids, expected[qi][0]) and np.array_equal(dist, expected[qi][1])
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    def mutator():
        try:
            i = 0
            while not stop.is_set() and i < 15:
                extra = util.gaussian_rows(500, d, seed=500 + i)
                st.insert_rows(np.arange(n + i * 500, n + (i + 1) * 500, dtype=np.uint64), extra)
                st.delete(n - 1 - i)  # rows far from every planted winner
                i += 1
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=searcher, args=(qi,)) for qi in range(8)] + [threading.Thread(target=mutator)]
    for t in threads:
        t.start()
    for t in threads[:-1]:
        t.join()
    stop.set()
    threads[-1].join()
    assert not errors, errors
    assert st.info()["num_rows"] > n
    st.close()


def test_experimental_fused_topk_path_is_exact():
    """Option "fused_topk" = 1 (per-wave register top-k inside K1; by default only for one query over a large corpus) must
    return exactly what the default dense path returns."""
    n, d = 30_000, 768
    corpus = util.gaussian_rows(n, d, seed=301)
    corpus[100:104] = corpus[7]  # exact ties
    queries = util.gaussian_rows(3, d, seed=302)
    base_ctx = oa.Context(0)
    st = make_store(base_ctx, corpus)
    want = [st.storage_search(queries, k) for k in (1, 10, 100, 128)]
    st.close()
    base_ctx.close()
    fctx = oa.Context(0)
    fctx.set_option("fused_topk", 1)
    st = make_store(fctx, corpus)
    for k, w in zip((1, 10, 100, 128), want):
        got = st.storage_search(queries, k)
        assert np.array_equal(got[0], w[0]) and np.array_equal(got[1], w[1]) and np.array_equal(got[2], w[2])
    st.close()
    fctx.close()


@pytest.mark.parametrize("d,metric", [(768, 0), (384, 0), (1024, 0), (100, 0), (256, 1), (768, 1)])
def test_batched_fp32_pass_equals_solo_queries(ctx, d, metric):
    """K1b (2..8 queries share one corpus pass) must return, per query, exactly what the single-query K1 pass
    returns: same ids, bit-identical distances — with tombstones and an allow-bitmap too."""
    n = 3001
    corpus = util.gaussian_rows(n, d, seed=31 + d)
    st = make_store(ctx, corpus, metric=oa.METRIC_L2SQ if metric else oa.METRIC_COSINE,
                    row_doc=np.arange(n, dtype=np.uint64) * 2 + 5)
    for doc in (5, 7, 2005):
        st.delete(doc)
    bm = oa.AllowBitmap.from_mask((np.arange(2 * n + 6) % 5) != 0)
    queries = util.gaussian_rows(19, d, seed=77 + d)
    for allow in (None, bm):
        solo = [st.storage_search(queries[i], 40, allow) for i in range(19)]
        for nq in (2, 3, 4, 5, 8, 9, 19):
            ids, dist, cnt = st.storage_search(queries[:nq], 40, allow)
            for i in range(nq):
                assert cnt[i] == solo[i][2][0]
                assert np.array_equal(ids[i], solo[i][0][0]), (nq, i)
                assert np.array_equal(dist[i].view(np.uint32), solo[i][1][0].view(np.uint32)), (nq, i)
    st.close()


def test_searches_are_not_blocked_by_a_large_ingest(ctx):
    """insert(&self) runs beside searches in the reference (index/mod.rs:1436, 1688-1698).  Here: 1 M rows are appended
    in slabs while another thread searches.  Searches must (a) never fail, (b) always return a consistent snapshot —
    sorted, k hits, every id below the row count published when the call returned — (c) see a planted perfect match
    as soon as its slab is published and never before, and (d) keep their latency: p95 during the ingest within 3x of
    the idle p95 (round 1 held an exclusive lock across each slab's H2D + sync: a search then waited for whole
    slabs, tens of ms).  No capacity is reserved up front: the arrays grow several times during the test (copied
    beside the running searches; readers only wait for the pointer swap)."""
    import threading
    import time

    d, n0, n_add, slab = 384, 400_000, 1_000_000, 50_000
    st = oa.EmbeddingFieldStorage(ctx, dimensions=d)          # no reserve_rows: growth happens during the test
    st.fill_synthetic(n0, seed=7)
    q = util.gaussian_rows(1, d, seed=8)[0]
    k = 20
    for _ in range(5):
        st.storage_search(q, k)
    idle = []
    for _ in range(60):
        t0 = time.perf_counter()
        st.storage_search(q, k)
        idle.append(time.perf_counter() - t0)
    errors, during, seen_plant = [], [], []
    done = threading.Event()
    plant_slab = 12                                             # the slab that carries the perfect match
    plant_id = 10_000_000

    def searcher():
        try:
            while not done.is_set():
                t0 = time.perf_counter()
                ids, dist, cnt = st.storage_search(q, k)
                during.append(time.perf_counter() - t0)
                rows_after = st.info()["num_rows"]
                assert cnt[0] == k and np.all(np.diff(dist[0]) >= 0)
                assert np.all((ids[0] < rows_after) | (ids[0] == plant_id))
                seen_plant.append((bool(ids[0, 0] == plant_id), rows_after))
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    th = threading.Thread(target=searcher)
    th.start()
    published_at = n0 + plant_slab * slab + 124  # the plant is row 123 of its slab (the library publishes sub-slabs)
    for i in range(n_add // slab):
        rows = util.gaussian_rows(slab, d, seed=1000 + i)
        ids = np.arange(n0 + i * slab, n0 + (i + 1) * slab, dtype=np.uint64)
        if i == plant_slab:
            rows[123] = q * np.float32(2.0)
            ids[123] = plant_id
        assert st.insert_rows(ids, rows) == slab
    done.set()
    th.join()
    assert not errors, errors
    assert st.info()["num_rows"] == n0 + n_add and len(during) > 20
    # (c) the plant shows up exactly from its slab's publication on
    for saw, rows_after in seen_plant:
        if saw:
            assert rows_after >= published_at
    assert any(s for s, _ in seen_plant) and st.storage_search(q, k)[0][0, 0] == plant_id
    # (d) latency
    p95_idle, p95_during = np.percentile(idle, 95), np.percentile(during, 95)
    assert p95_during <= 3.0 * p95_idle + 2e-3, (p95_idle, p95_during)
    # deletes by a device pass over row_doc, compaction re-packs; searches still exact
    st.delete(plant_id)
    assert st.info()["pending_ops"] == 1 and st.storage_search(q, k)[0][0, 0] != plant_id
    before = st.storage_search(q, k)
    st.compact(3)
    after = st.storage_search(q, k)
    assert st.info()["pending_ops"] == 0 and st.info()["num_rows"] == n0 + n_add - 1
    assert np.array_equal(before[0], after[0]) and np.allclose(before[1], after[1], atol=1e-6)
    st.close()


def test_ties_at_the_cut_are_decided_by_row_index_whatever_the_list_length(ctx):
    """DESIGN §3 rule 4 (the oracle's orc_vector_search): among rows at the SAME distance the k-th boundary keeps the lowest
    ROW indices; the survivors are then ordered by (distance, DocumentId, row).  With DocumentIds that are not monotonic in
    the row index the two orders differ — and round 3's final selection cut short lists (<= 2 x pow2(k) candidates) by
    DocumentId and longer ones by row index.  Exact duplicates of one row make the ties; three store sizes put
    the final selection on its short, its medium and its reduced (multi-chunk) path."""
    d, k = 64, 100
    rng = np.random.default_rng(77)
    base = rng.standard_normal(d).astype(np.float32)
    q = base + 0.01 * rng.standard_normal(d).astype(np.float32)
    for n in (180, 1000, 40_000):
        rows = rng.standard_normal((n, d)).astype(np.float32)
        dup = np.sort(rng.choice(n, size=150, replace=False))  # 150 rows at the same (smallest) distance, k = 100 of them survive
        rows[dup] = base
        doc_ids = (np.uint64(10 * n) - np.arange(n, dtype=np.uint64) * np.uint64(7)).astype(np.uint64)  # descending: lowest rows = highest ids
        st = oa.EmbeddingFieldStorage(ctx, dimensions=d)
        st.insert_rows(doc_ids, rows)
        ids, dist, cnt = st.storage_search(q, k)
        o_ids, o_dist, o_rows = orc.vector_search(rows, doc_ids, q, k)
        assert cnt[0] == k
        assert set(o_rows.tolist()) == set(dup[:k].tolist())  # the oracle's rule: the lowest row indices among the ties
        assert ids[0].tolist() == o_ids.tolist(), n
        assert np.max(np.abs(dist[0] - o_dist)) <= 1e-6
        st.close()


@pytest.mark.parametrize("mode", ["1", "2", "0"])
def test_a_lone_query_over_a_long_dense_list_takes_one_round_per_part(mode):
    """Round 5: the distance array of a lone query (> 16 chunks of 8 192 values) is cut by
    pairs_reduce_wide_kernel — a part's <= 32 768 values in registers, one bound, one cut.  Option "select_wide" = 2 forces the
    rounds it falls back to when more keys reach the bound than LDS holds, 0 the rounds of round 4: three forms, one answer —
    checked against a host selection over the distances the device reports for every row (limit = n is not offered: the
    distances come from the rows read back), with exact ties across the cut and NaN-free inputs of three lengths."""
    c = oa.Context(0)
    c.set_option("select_wide", int(mode))
    d = 16
    rng = np.random.default_rng(5150)
    for n, ks in ((140_000, (1, 100)), (300_000, (10, 100, 256)), (1_000_003, (100, 128))):
        rows = rng.standard_normal((n, d)).astype(np.float32)
        q = rng.standard_normal(d).astype(np.float32)
        dup = np.sort(rng.choice(n, size=300, replace=False))
        rows[dup] = q * np.float32(2.0)  # 300 rows at distance ~0: the cut falls among exact ties for every k here
        st = make_store(c, rows)
        o_d = orc.distances(rows, q)
        for k in ks:
            ids, dist, cnt = st.storage_search(q, k)
            assert cnt[0] == k
            assert ids[0].tolist() == dup[:k].tolist(), (n, k)  # ties: lowest rows, then DocumentId order (= row order here)
            assert np.max(np.abs(dist[0] - o_d[ids[0].astype(np.int64)])) <= TOL
        # without the ties: the k best of a plain gaussian corpus
        rows[dup] = rng.standard_normal((300, d)).astype(np.float32)
        st.close()
        st = make_store(c, rows)
        o_d = orc.distances(rows, q)
        for k in ks:
            ids, dist, cnt = st.storage_search(q, k)
            order = np.lexsort((np.arange(n), o_d))[: k + 8]
            assert cnt[0] == k and np.all(np.diff(dist[0]) >= 0)
            assert np.max(np.abs(dist[0] - o_d[order[:k]])) <= TOL
            # ids identical wherever the distances are separated by more than the fp32 summation-order noise
            sep = np.abs(np.diff(o_d[order])) > 1e-5
            for j in range(k):
                if (j == 0 or sep[j - 1]) and sep[j]:
                    assert ids[0, j] == order[j], (n, k, j)
        st.close()
    c.close()
