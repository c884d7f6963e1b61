"""GPU parity of K1m: batches of >= 9 queries over the PLAIN fp32 store on the matrix cores (v_mfma_f32_32x32x2_f32,
csrc/vec_f32_mfma.hip) — north_star's "MFMA-backed batched-query x corpus GEMM when Q>1" for the reference's own dtype
(embedding_field.rs:66,88,250-278: Vec<f32> rows, cosine).

The bar: distances within 1e-4 of the oracle evaluated on the stored rows, id sets equal under the tie-aware checker
(util.assert_topk_sound) — and the SAME BITS as the single-query scan K1, whatever batch a query was asked in: K1m proposes
k + spare candidates, K1's own arithmetic decides, an unproven list (ties around the k-th place) is re-answered by K1 on the
device (vec_store.hip search_enqueue_f32_batch).
"""
import numpy as np
import pytest

import oramacore_amd as oa
import util
from oracle import oracle as orc

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(params=["K1x", "K1m"], autouse=True)
def plan(request, ctx):
    """Every case runs under both candidate scans: K1x (default where every row has a sound fp16 image: the fp32 rows rounded to
    fp16 in registers, <= 64 queries per pass) and K1m (f32 x f32 on the matrix cores, <= 32 per pass)."""
    ctx.set_option("f32_batch_cvt", 1 if request.param == "K1x" else 0)
    yield request.param
    ctx.set_option("f32_batch_cvt", 1)


def make_store(ctx, corpus, row_doc=None):
    n, d = corpus.shape
    st = oa.EmbeddingFieldStorage(ctx, dimensions=d)
    ids = np.arange(n, dtype=np.uint64) if row_doc is None else row_doc
    assert st.insert_rows(ids, corpus) == n
    return st


def check(st, corpus, queries, k, allow=None, dead_rows=None, what=""):
    ids, dist, cnt = st.storage_search(queries, k, allow)
    for qi in range(queries.shape[0]):
        full = orc.distances(corpus, queries[qi]).astype(np.float64)
        if dead_rows is not None:
            full[dead_rows] = np.nan
        if allow is not None:
            mask = np.array([allow.contains(int(d)) for d in range(corpus.shape[0])])
            full[~mask] = np.nan
        m = int(cnt[qi])
        util.assert_topk_sound(ids[qi, :m], dist[qi, :m], full, k, TOL, f"{what} q{qi}")
    return ids, dist, cnt


@pytest.mark.parametrize("d", [32, 384, 768, 864, 1024])
@pytest.mark.parametrize("nq", [9, 17, 32, 33, 70, 130])
def test_head_only_batches(ctx, d, nq):
    """N below the dense head, a ragged last tile (n % 32 != 0): one and several passes, ragged query tiles.  (1 024 dimensions:
    K1x takes 32 queries per pass there, K1m's query tile does not fit LDS — K1 / K1b answer under that plan.)"""
    n = 3000 + d + 7
    corpus = util.gaussian_rows(n, d, seed=d)
    queries = util.gaussian_rows(nq, d, seed=d + nq)
    st = make_store(ctx, corpus)
    check(st, corpus, queries, 100, what=f"d={d} nq={nq}")
    st.close()


def test_filter_path_random_and_adversarial_order(ctx):
    """N above the dense head (131072 rows): the rest goes through the threshold filter; then a corpus ordered by INCREASING
    similarity to query 0 (every later row beats the running threshold: the candidate lists' worst case)."""
    n, d, k = 200_003, 384, 100
    corpus = util.gaussian_rows(n, d, seed=7)
    queries = util.gaussian_rows(32, d, seed=8)
    st = make_store(ctx, corpus)
    check(st, corpus, queries, k, what="random order")
    check(st, corpus, queries[:9], 7, what="random order small k")
    st.close()
    sim = corpus @ queries[0] / np.linalg.norm(corpus, axis=1)
    order = np.argsort(sim, kind="stable")
    corpus2 = np.ascontiguousarray(corpus[order])
    st = make_store(ctx, corpus2)
    check(st, corpus2, queries[:9], k, what="adversarial order")
    st.close()


def test_same_bits_in_every_batch(ctx):
    """One fixed fmaf chain per (row, query): a query's answer is the same bits alone in a batch of 9 (padded by other
    queries), in a full tile of 32, across a pass boundary (position 40 of 70) — ids, distances and counts."""
    n, d, k = 150_000, 768, 100
    corpus = util.gaussian_rows(n, d, seed=31)
    queries = util.gaussian_rows(70, d, seed=32)
    st = make_store(ctx, corpus)
    i70, d70, c70 = st.storage_search(queries, k)
    i32, d32, c32 = st.storage_search(queries[:32], k)
    assert np.array_equal(i70[:32], i32) and np.array_equal(d70[:32].view(np.uint32), d32.view(np.uint32)) and np.array_equal(c70[:32], c32)
    sel = np.array([40, 3, 69, 11, 12, 13, 14, 15, 16])
    i9, d9, c9 = st.storage_search(queries[sel], k)
    assert np.array_equal(i9, i70[sel]) and np.array_equal(d9.view(np.uint32), d70[sel].view(np.uint32)) and np.array_equal(c9, c70[sel])
    # ... and the bits K1 / K1b answer with when the matrix path is off
    st.ctx.set_f32_batch(0)
    try:
        ik, dk, ck = st.storage_search(queries[:16], k)
        i1, d1, c1 = st.storage_search(queries[40], k)
    finally:
        st.ctx.set_f32_batch(9)
    assert np.array_equal(ck, c70[:16]) and np.array_equal(ik, i70[:16]) and np.array_equal(dk.view(np.uint32), d70[:16].view(np.uint32))
    assert np.array_equal(i1[0], i70[40]) and np.array_equal(d1[0].view(np.uint32), d70[40].view(np.uint32)) and c1[0] == c70[40]
    st.close()


def test_deletes_filter_and_compaction(ctx):
    n, d, k = 140_000, 384, 50
    corpus = util.gaussian_rows(n, d, seed=17)
    queries = util.gaussian_rows(12, d, seed=18)
    st = make_store(ctx, corpus)
    dead = np.array([3, 64, 65, 131071, 131072, 139_999], dtype=np.int64)
    for r in dead:
        st.delete(int(r))
    allow = oa.AllowBitmap.from_mask(np.arange(n) % 3 != 0)
    check(st, corpus, queries, k, dead_rows=dead, what="dead")
    check(st, corpus, queries, k, allow=allow, dead_rows=dead, what="dead+filter")
    before = st.storage_search(queries, k)
    st.compact(3)
    assert st.info()["num_rows"] == n - len(dead) and not st.has_pending_ops()
    after = st.storage_search(queries, k)
    assert np.array_equal(before[0], after[0]) and np.array_equal(before[1].view(np.uint32), after[1].view(np.uint32))
    st.close()


def test_growth_and_appends_keep_the_tail_tile_right(ctx):
    """Rows appended in odd-sized blocks (the store grows in place; a partial last tile is read whole and masked): every prefix
    answers like a fresh store of the same rows."""
    d, k = 384, 20
    corpus = util.gaussian_rows(9000, d, seed=41)
    queries = util.gaussian_rows(10, d, seed=42)
    st = oa.EmbeddingFieldStorage(ctx, dimensions=d)
    at = 0
    for step in (1, 30, 33, 1000, 4097, 3839):
        st.insert_rows(np.arange(at, at + step, dtype=np.uint64), corpus[at:at + step])
        at += step
        check(st, corpus[:at], queries, k, what=f"rows={at}")
    st.close()


def test_unsupported_shapes_keep_the_valu_path(ctx):
    """L2 stores, dimensions that are not whole chunks or whose query tile does not fit LDS: K1 / K1b answer as before."""
    for d, metric in ((100, None), (1024, None), (384, "l2")):
        n = 5000
        corpus = util.gaussian_rows(n, d, seed=d)
        queries = util.gaussian_rows(12, d, seed=d + 1)
        kw = {"metric": oa.METRIC_L2SQ} if metric else {}
        st = oa.EmbeddingFieldStorage(ctx, dimensions=d, **kw)
        st.insert_rows(np.arange(n, dtype=np.uint64), corpus)
        ids, dist, cnt = st.storage_search(queries, 10)
        for j in range(12):
            full = (orc.distances(corpus, queries[j]) if not metric else
                    ((corpus.astype(np.float64) - queries[j].astype(np.float64)) ** 2).sum(axis=1)).astype(np.float64)
            util.assert_topk_sound(ids[j, :cnt[j]], dist[j, :cnt[j]], full, 10, 1e-3 if metric else TOL, f"d={d} {metric} q{j}")
        st.close()


def test_duplicates_around_the_kth_place_take_the_fallback_and_stay_exact(ctx):
    """Rows that tie with the k-th best (copies of one row, more of them than the spare candidates): the completeness proof of
    the candidate list fails, the query is re-answered by K1 on the device — same bits as the solo scan, ties by DocumentId."""
    n, d, k = 40_000, 384, 10
    corpus = util.gaussian_rows(n, d, seed=51)
    queries = util.gaussian_rows(12, d, seed=52)
    target = (queries[3] + 0.3 * util.gaussian_rows(1, d, seed=53)[0]).astype(np.float32)
    dup = np.random.default_rng(54).choice(n, size=120, replace=False)
    corpus[dup] = target  # 120 identical rows near query 3: the 10th .. 120th best tie exactly
    st = make_store(ctx, corpus)
    ib, db, cb = st.storage_search(queries, k)
    st.ctx.set_f32_batch(0)
    try:
        for j in range(12):
            i1, d1, c1 = st.storage_search(queries[j], k)
            assert np.array_equal(i1[0], ib[j]) and np.array_equal(d1[0].view(np.uint32), db[j].view(np.uint32)) and c1[0] == cb[j], j
    finally:
        st.ctx.set_f32_batch(9)
    assert set(ib[3].tolist()) == set(sorted(dup.tolist())[:k])  # the tie rule: lower DocumentId first
    # a zero query: every distance is 1.0, every list ties — the fallback again
    z = np.zeros((9, d), dtype=np.float32)
    iz, dz, cz = st.storage_search(z, k)
    assert np.all(dz == 1.0) and iz[0].tolist() == list(range(k)) and np.all(cz == k)
    st.close()


def test_the_plan_follows_the_rows_and_the_queries(ctx, plan):
    """K1x needs a sound fp16 image of every row (|x_i| < 6e4, |x|^2 >= 1e-4): a store that ever accepted another row proposes with
    K1m from then on; a QUERY without one (tiny norm, an element beyond the fp16 range) flags itself on the device and is answered
    by K1.  Answers equal the solo scan's bits in every case."""
    n, d, k = 30_000, 384, 10
    corpus = util.gaussian_rows(n, d, seed=61)
    queries = util.gaussian_rows(12, d, seed=62)
    queries[2] *= np.float32(1e-4)      # |q|^2 ~ 4e-6: no usable fp16 image
    queries[5][7] = np.float32(7.0e4)   # beyond the fp16 range
    st = make_store(ctx, corpus)

    def launches():
        ctx.prof_reset(); ctx.prof_enable(True)
        out = st.storage_search(queries, k)
        ctx.prof_enable(False)
        return out, ctx.prof_get("vec_scan_f32_cvt")[1], ctx.prof_get("vec_scan_f32_mfma")[1]

    (ib, db, cb), n_cvt, n_mfma = launches()
    assert (n_cvt > 0 and n_mfma == 0) if plan == "K1x" else (n_cvt == 0 and n_mfma > 0)
    ctx.set_f32_batch(0)
    try:
        for j in range(12):
            i1, d1, c1 = st.storage_search(queries[j], k)
            assert np.array_equal(i1[0], ib[j]) and np.array_equal(d1[0].view(np.uint32), db[j].view(np.uint32)) and c1[0] == cb[j], j
    finally:
        ctx.set_f32_batch(9)
    # one row beyond the fp16 range: the store leaves K1x for good
    big = util.gaussian_rows(1, d, seed=63)
    big[0][3] = np.float32(1.0e5)
    st.insert_rows(np.array([n], dtype=np.uint64), big)
    corpus2 = np.concatenate([corpus, big])
    (ib, db, cb), n_cvt, n_mfma = launches()
    assert n_cvt == 0 and n_mfma > 0
    for j in (0, 1, 3, 4):
        full = orc.distances(corpus2, queries[j]).astype(np.float64)
        util.assert_topk_sound(ib[j, :cb[j]], db[j, :cb[j]], full, k, TOL, f"after the unsafe row q{j}")
    st.close()


@pytest.mark.parametrize("seed", range(10))
def test_random_shapes_equal_the_plain_path(ctx, seed):
    """Random dimensions, row counts (down to fewer rows than k, than a tile), batch sizes, k, tombstones and filters: the matrix
    path's answers are the plain path's (K1 / K1b, the matrix path switched off) bit for bit — ids, distances, counts."""
    rng = np.random.default_rng(1000 + seed)
    d = int(rng.choice([32, 64, 96, 128, 256, 384, 512, 768, 1024]))
    n = [1, 7, 31, 33, 100, 5000, 40_000, 140_000, 270_001, 200_000][seed]  # (every size class once: a random draw clusters)
    nq = int(rng.choice([9, 10, 31, 32, 33, 64, 65, 100, 129]))
    k = int(rng.choice([1, 5, 10, 100, 128]))
    corpus = util.gaussian_rows(n, d, seed=2000 + seed)
    if n > 50 and seed % 3 == 0:
        corpus[rng.choice(n, size=min(n // 2, 300), replace=False)] = corpus[0]  # a crowd of duplicates
    queries = util.gaussian_rows(nq, d, seed=3000 + seed)
    if seed % 4 == 1:
        queries[0] = 0.0  # a zero query
    row_doc = (np.arange(n, dtype=np.uint64) * 3 + 5) if seed % 2 else None
    st = make_store(ctx, corpus, row_doc)
    docs = np.arange(n, dtype=np.uint64) if row_doc is None else row_doc
    if n > 40:
        for r in rng.choice(n, size=5, replace=False):
            st.delete(int(docs[r]))
    allow = None
    if seed % 3 == 1:
        allow = oa.AllowBitmap(int(docs.max()) + 1, docs[rng.random(n) < 0.7])
    got = st.storage_search(queries, k, allow)
    ctx.set_f32_batch(0)
    try:
        want = st.storage_search(queries, k, allow)
    finally:
        ctx.set_f32_batch(9)
    assert np.array_equal(got[2], want[2]), (d, n, nq, k)
    for j in range(nq):
        m = int(want[2][j])
        assert np.array_equal(got[0][j, :m], want[0][j, :m]), (d, n, nq, k, j)
        assert np.array_equal(got[1][j, :m].view(np.uint32), want[1][j, :m].view(np.uint32)), (d, n, nq, k, j)
    st.close()
