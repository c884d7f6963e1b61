"""Two-stage exact search (ORAMA_DTYPE_F32_SHADOW16: fp32 rows + fp16 shadow): the answers must be the plain fp32
scan's bit for bit — ids, distances, counts — for single queries and batches, under filters, deletes, live inserts and
compaction; adversarial data (thousands of near-duplicates around the k-th distance) must take the fp32 fallback and
still agree; rows without an fp16 error bound switch the plan off."""
import numpy as np
import pytest

import oramacore_amd as oa
from oracle import oracle as orc  # checker only

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = oa.Context(0)
    c.set_two_stage(True, always=True)  # the stores here are far below the size where the plan is chosen by itself
    yield c
    c.close()


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def pair(ctx, dim, reserve=0):
    plain = oa.EmbeddingFieldStorage(ctx, dimensions=dim, dtype=oa.DTYPE_F32, reserve_rows=reserve)
    shadow = oa.EmbeddingFieldStorage(ctx, dimensions=dim, dtype=oa.DTYPE_F32_SHADOW16, reserve_rows=reserve)
    return plain, shadow


def same(plain, shadow, queries, k, allow=None, tag=""):
    pi, pd, pn = plain.storage_search(queries, k, allow)
    si, sd, sn = shadow.storage_search(queries, k, allow)
    assert pn.tolist() == sn.tolist(), tag
    for j in range(len(pn)):
        n = int(pn[j])
        assert pi[j, :n].tolist() == si[j, :n].tolist(), (tag, j)
        assert np.array_equal(bits(pd[j, :n]), bits(sd[j, :n])), (tag, j)
    return si, sd, sn


@pytest.mark.parametrize("dim", [64, 384, 768, 1000])
def test_equals_the_fp32_scan(ctx, dim):
    rng = np.random.default_rng(dim)
    n = 60_000
    rows = (rng.standard_normal((n, dim)) * rng.uniform(0.5, 2.0, size=(n, 1))).astype(np.float32)
    ids = (np.arange(n, dtype=np.uint64) * np.uint64(3) + np.uint64(7))
    ids[1000:1010] = ids[999]  # several rows of one document
    plain, shadow = pair(ctx, dim)
    assert plain.insert_rows(ids, rows) == shadow.insert_rows(ids, rows) == n
    q = rng.standard_normal((70, dim)).astype(np.float32)
    for k in (1, 10, 100, 1000):
        same(plain, shadow, q[:1], k, tag=("q1", k))
        same(plain, shadow, q[:9], k, tag=("q9", k))
    same(plain, shadow, q, 100, tag="q70")
    # the oracle agrees (distance within the declared 1e-4, ids wherever distances are separated)
    si, sd, sn = shadow.storage_search(q[:3], 50)
    for j in range(3):
        od = orc.distances(rows, q[j])
        order = np.lexsort((np.arange(n), ids, od))[:50]
        assert np.max(np.abs(od[order] - sd[j])) <= 1e-4
    # filter + deletes + inserts + compaction
    allow_ids = ids[rng.random(n) < 0.5]
    bm = oa.AllowBitmap(int(ids.max()) + 1, allow_ids)
    same(plain, shadow, q[:5], 100, bm, tag="filter")
    for d in rng.choice(ids, size=300, replace=False):
        plain.delete(int(d))
        shadow.delete(int(d))
    same(plain, shadow, q[:5], 100, tag="deleted")
    same(plain, shadow, q[:5], 100, bm, tag="deleted+filter")
    more = rng.standard_normal((5000, dim)).astype(np.float32)
    more_ids = np.arange(5000, dtype=np.uint64) + np.uint64(10**9)
    plain.insert_rows(more_ids, more)
    shadow.insert_rows(more_ids, more)
    same(plain, shadow, q[:5], 100, tag="inserted")
    plain.compact(2)
    shadow.compact(2)
    assert plain.info()["num_rows"] == shadow.info()["num_rows"]
    same(plain, shadow, q[:9], 100, tag="compacted")
    info = shadow.info()
    assert info["two_stage_queries"] > 0
    # the plan switched off gives the same answers through the same store
    ctx.set_two_stage(False)
    before = shadow.info()["two_stage_queries"]
    same(plain, shadow, q[:3], 10, tag="plan off")
    assert shadow.info()["two_stage_queries"] == before
    ctx.set_two_stage(True, always=True)
    plain.close()
    shadow.close()


def test_near_duplicates_take_the_fallback(ctx):
    """6 000 rows within ~1e-6 of the same direction: the band around the k-th shadow distance holds more rows than the
    candidate list — the query must be answered by the fp32 scan (and counted as a fallback), not by an incomplete list."""
    rng = np.random.default_rng(5)
    dim, n = 256, 40_000
    rows = rng.standard_normal((n, dim)).astype(np.float32)
    centre = rng.standard_normal(dim).astype(np.float32)
    rows[:6000] = centre + (rng.standard_normal((6000, dim)) * 1e-4).astype(np.float32)
    ids = np.arange(n, dtype=np.uint64)
    plain, shadow = pair(ctx, dim)
    plain.insert_rows(ids, rows)
    shadow.insert_rows(ids, rows)
    q = np.stack([centre + (rng.standard_normal(dim) * 1e-3).astype(np.float32), rng.standard_normal(dim).astype(np.float32)])
    same(plain, shadow, q, 100, tag="duplicates")
    info = shadow.info()
    assert info["two_stage_fallbacks"] >= 1 and info["two_stage_queries"] >= 2
    # exact duplicates: ties resolved by (doc, row) like the fp32 path
    rows2 = np.repeat(rng.standard_normal((50, dim)).astype(np.float32), 40, axis=0)
    p2, s2 = pair(ctx, dim)
    p2.insert_rows(np.arange(2000, dtype=np.uint64)[::-1].copy(), rows2)
    s2.insert_rows(np.arange(2000, dtype=np.uint64)[::-1].copy(), rows2)
    same(p2, s2, rows2[:3] + np.float32(0.01), 90, tag="ties")
    for st in (plain, shadow, p2, s2):
        st.close()


def test_rows_without_an_error_bound_switch_the_plan_off(ctx):
    rng = np.random.default_rng(9)
    dim = 128
    rows = rng.standard_normal((5000, dim)).astype(np.float32)
    plain, shadow = pair(ctx, dim)
    ids = np.arange(5000, dtype=np.uint64)
    plain.insert_rows(ids, rows)
    shadow.insert_rows(ids, rows)
    q = rng.standard_normal((2, dim)).astype(np.float32)
    same(plain, shadow, q, 10)
    used = shadow.info()["two_stage_queries"]
    assert used == 2
    tiny = (rng.standard_normal((1, dim)) * 1e-5).astype(np.float32)  # norm ~1e-4: fp16 subnormals
    plain.insert_rows(np.array([10**6], dtype=np.uint64), tiny)
    shadow.insert_rows(np.array([10**6], dtype=np.uint64), tiny)
    same(plain, shadow, q, 10)
    same(plain, shadow, tiny * 3, 10)
    assert shadow.info()["two_stage_queries"] == used  # the plain scan answered
    # a query outside the fp16 range is answered by the plain scan as well (another store, still two-stage capable)
    p2, s2 = pair(ctx, dim)
    p2.insert_rows(ids, rows)
    s2.insert_rows(ids, rows)
    same(p2, s2, q * np.float32(1e5), 10)
    assert s2.info()["two_stage_queries"] == 0
    same(p2, s2, q, 10)
    assert s2.info()["two_stage_queries"] == 2
    for st in (plain, shadow, p2, s2):
        st.close()


def test_batcher_on_a_shadow_store(ctx):
    import threading

    rng = np.random.default_rng(13)
    dim, n = 384, 50_000
    rows = rng.standard_normal((n, dim)).astype(np.float32)
    ids = np.arange(n, dtype=np.uint64)
    plain, shadow = pair(ctx, dim)
    plain.insert_rows(ids, rows)
    shadow.insert_rows(ids, rows)
    qs = rng.standard_normal((48, dim)).astype(np.float32)
    batcher = oa.SearchBatcher(shadow, max_batch=64)
    got = [None] * len(qs)

    def worker(t):
        for i in range(t, len(qs), 12):
            got[i] = batcher.search(qs[i], 20)

    ths = [threading.Thread(target=worker, args=(t,)) for t in range(12)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    pi, pd, pn = plain.storage_search(qs, 20)
    for i in range(len(qs)):
        assert got[i][0].tolist() == pi[i].tolist() and np.array_equal(bits(got[i][1]), bits(pd[i]))
    batcher.close()
    plain.close()
    shadow.close()


def test_hybrid_search_on_a_shadow_store(ctx):
    """orama_hybrid_search takes the two-stage plan for its vector leg: same (ids, scores, count) as with a plain store."""
    from oramacore_amd import fulltext as ft

    rng = np.random.default_rng(17)
    dim, n = 256, 30_000
    rows = rng.standard_normal((n, dim)).astype(np.float32)
    ids = np.arange(n, dtype=np.uint64)
    plain, shadow = pair(ctx, dim)
    plain.insert_rows(ids, rows)
    shadow.insert_rows(ids, rows)
    lists = []
    for l in range(6):
        local = np.sort(rng.choice(n, size=int(rng.integers(200, 4000)), replace=False))
        lists.append(ft.PostingList(field=0, docs=ids[local], tf=rng.integers(1, 5, size=len(local)),
                                    field_len=rng.integers(5, 200, size=len(local))))
    post = ft.PostingsStore(ctx)
    post.build(ids, [60.0], lists)
    before = shadow.info()["two_stage_queries"]
    for i in range(6):
        q = rng.standard_normal(dim).astype(np.float32)
        refs = [(t, int(l), 1.0) for t, l in enumerate(rng.choice(6, size=3, replace=False))]
        a = post.hybrid_search(plain, q, 50, 0.0, refs, 3, float(n), 40)
        b = post.hybrid_search(shadow, q, 50, 0.0, refs, 3, float(n), 40)
        assert a[2] == b[2] and a[0].tolist() == b[0].tolist() and np.array_equal(bits(a[1]), bits(b[1])), i
    assert shadow.info()["two_stage_queries"] == before + 6
    post.close()
    plain.close()
    shadow.close()


def test_tiny_and_growing_stores(ctx):
    """Fewer rows than candidates asked for; a store that grows past its reservation while searches run."""
    import threading

    rng = np.random.default_rng(21)
    dim = 128
    plain, shadow = pair(ctx, dim, reserve=64)
    rows = rng.standard_normal((50, dim)).astype(np.float32)
    ids = np.arange(50, dtype=np.uint64)
    plain.insert_rows(ids, rows)
    shadow.insert_rows(ids, rows)
    q = rng.standard_normal((3, dim)).astype(np.float32)
    for k in (1, 10, 50, 200):
        same(plain, shadow, q, k, tag=("tiny", k))
    # concurrent ingest: every answer must be sorted, complete and made of published documents only
    stop = threading.Event()
    errors = []

    def searcher():
        try:
            while not stop.is_set():
                i, d, c = shadow.storage_search(q[:1], 20)
                assert c[0] == 20 and np.all(np.diff(d[0]) >= 0)
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    ths = [threading.Thread(target=searcher) for _ in range(3)]
    for t in ths:
        t.start()
    base = 50
    for slab in range(12):
        more = rng.standard_normal((20_000, dim)).astype(np.float32)
        mids = np.arange(base, base + 20_000, dtype=np.uint64)
        shadow.insert_rows(mids, more)
        plain.insert_rows(mids, more)
        base += 20_000
    stop.set()
    for t in ths:
        t.join()
    assert not errors, errors
    same(plain, shadow, q, 100, tag="after the ingest")
    assert shadow.info()["num_rows"] == base
    plain.close()
    shadow.close()


def test_the_plan_is_chosen_only_where_it_pays(ctx):
    """Default mode: a small store answers up to 8 queries by the plain scan (cheaper than the second stage's launches)
    and larger batches by the two stages."""
    rng = np.random.default_rng(23)
    dim, n = 128, 20_000
    rows = rng.standard_normal((n, dim)).astype(np.float32)
    plain, shadow = pair(ctx, dim)
    ids = np.arange(n, dtype=np.uint64)
    plain.insert_rows(ids, rows)
    shadow.insert_rows(ids, rows)
    q = rng.standard_normal((20, dim)).astype(np.float32)
    ctx.set_two_stage(True)
    try:
        same(plain, shadow, q[:8], 10)
        assert shadow.info()["two_stage_queries"] == 0
        same(plain, shadow, q, 10)
        assert shadow.info()["two_stage_queries"] == 20
    finally:
        ctx.set_two_stage(True, always=True)
    plain.close()
    shadow.close()


def test_the_device_path_decides_the_fallback_on_the_device(ctx):
    """orama_vec_search_device / _packed_device and the pipelined session on a shadow store: the two-stage plan
    with NO host between its stages and the consumer of the answers.  Queries that are not proven — a query in the middle of
    6 000 near-duplicates — and queries the fp16 shadow cannot serve at all (a zero vector, components beyond the fp16
    range, a NaN) are re-answered on the device by K1's own kernel over a device-made list of queries: every answer equals
    the plain fp32 store's bit for bit, under a filter too; k beyond the fused scan's 128 keeps the fp32 scan."""
    from oramacore_amd import _native as N
    from oramacore_amd.shard_group import ShardGroup

    lib = N.load()
    rng = np.random.default_rng(17)
    dim, n, k = 256, 40_000, 100
    rows = rng.standard_normal((n, dim)).astype(np.float32)
    centre = rng.standard_normal(dim).astype(np.float32)
    rows[:6000] = centre + (rng.standard_normal((6000, dim)) * 1e-4).astype(np.float32)
    ids = np.arange(n, dtype=np.uint64) * np.uint64(5) + np.uint64(3)
    plain, shadow = pair(ctx, dim)
    plain.insert_rows(ids, rows)
    shadow.insert_rows(ids, rows)
    qs = rng.standard_normal((16, dim)).astype(np.float32)
    qs[1] = centre + (rng.standard_normal(dim) * 1e-3).astype(np.float32)  # not proven: the band holds thousands of rows
    qs[4] = 0.0                                                             # |q| = 0
    qs[7] *= np.float32(1e5)                                                # beyond the fp16 range
    qs[9, 3] = np.nan
    qs[12] = centre
    Q = qs.shape[0]
    e_ids, e_dist, e_cnt = plain.storage_search(qs, k)

    def device_search(store, q, kk, allow=None):
        d_q = oa.DeviceBuffer(ctx, q.nbytes).upload(q)
        d_i, d_d, d_n = oa.DeviceBuffer(ctx, len(q) * kk * 8), oa.DeviceBuffer(ctx, len(q) * kk * 4), oa.DeviceBuffer(ctx, len(q) * 4)
        tok, nbits = allow.ffi_args() if allow is not None else (None, 0)
        N.check(lib.orama_vec_search_device(store.handle, d_q.ptr, len(q), kk, tok, nbits, d_i.ptr, d_d.ptr, d_n.ptr, None))
        ctx.synchronize()
        out = (d_i.download(np.uint64, len(q) * kk).reshape(len(q), kk), d_d.download(np.float32, len(q) * kk).reshape(len(q), kk),
               d_n.download(np.uint32, len(q)))
        for b in (d_q, d_i, d_d, d_n):
            b.free()
        return out

    def equal(got, exp, tag):
        gi, gd, gn = got
        xi, xd, xn = exp
        assert gn.tolist() == xn.tolist(), tag
        for j in range(len(gn)):
            m = int(gn[j])
            assert gi[j, :m].tolist() == xi[j, :m].tolist(), (tag, j)
            assert np.array_equal(bits(gd[j, :m]), bits(xd[j, :m])), (tag, j)

    before = shadow.info()["two_stage_queries"]
    equal(device_search(shadow, qs, k), (e_ids, e_dist, e_cnt), "batch of 16")
    assert shadow.info()["two_stage_queries"] == before + Q  # the plan ran (the device form counts its queries)
    for j in (0, 1, 4, 7, 9):
        equal(device_search(shadow, qs[j:j + 1], k), (e_ids[j:j + 1], e_dist[j:j + 1], e_cnt[j:j + 1]), ("solo", j))
    # a batch in which EVERY query is flagged
    allbad = np.zeros((5, dim), dtype=np.float32)
    equal(device_search(shadow, allbad, k), plain.storage_search(allbad, k), "all flagged")
    # more flagged queries than the fallback keeps list sets for (64; round 5): 150 of 200 — three rounds of the same launches
    many = rng.standard_normal((200, dim)).astype(np.float32)
    bad = np.sort(rng.choice(200, size=150, replace=False))
    many[bad[:50]] = 0.0
    many[bad[50:100]] *= np.float32(1e5)
    many[bad[100:]] = centre + (rng.standard_normal((50, dim)) * 1e-3).astype(np.float32)
    equal(device_search(shadow, many, k), plain.storage_search(many, k), "150 of 200 flagged")
    # under a filter (resident bitmap)
    mask_ids = ids[rng.random(n) < 0.4]
    bm = oa.AllowBitmap(int(ids.max()) + 1, mask_ids).to_device(ctx)
    host_bm = oa.AllowBitmap(int(ids.max()) + 1, mask_ids)
    equal(device_search(shadow, qs, 50, bm), plain.storage_search(qs, 50, host_bm), "filtered")
    bm.close()
    # k beyond the fused scan's lists: the fp32 scan answers, the plan's counter stands still
    before = shadow.info()["two_stage_queries"]
    equal(device_search(shadow, qs[:3], 200), plain.storage_search(qs[:3], 200), "k = 200")
    assert shadow.info()["two_stage_queries"] == before
    # the pipelined session (bench.py's loop): 4 queries per step, two slots in flight
    group = ShardGroup([0])  # (its own context: the stores of a session belong to the group's)
    group.ctx(0).set_two_stage(True, always=True)
    gp, gs = pair(group.ctx(0), dim)
    gp.insert_rows(ids, rows)
    gs.insert_rows(ids, rows)
    for force in (False, True):
        sess = group.session([gs], qs, 4, k, n_slots=2, force_exchange=force)
        for step in range(Q // 4):
            sess.step(step)
            if step % 2 == 1:  # both slots hold a finished step
                sess.sync()
                for slot, st in ((0, step - 1), (1, step)):
                    s_ids, s_dist, s_cnt = sess.result(slot)
                    equal((s_ids, s_dist, s_cnt), (e_ids[4 * st:4 * st + 4], e_dist[4 * st:4 * st + 4], e_cnt[4 * st:4 * st + 4]),
                          ("session", force, st))
        sess.close()
    assert gs.info()["two_stage_queries"] >= 2 * Q
    gp.close()
    gs.close()
    group.close()
    plain.close()
    shadow.close()
