"""orama_shard_* — the multi-GPU exchange inside the library, driven through the C ABI without torch.

What a 1-GPU box can check:
  * co-located groups (G shards on GPU 0): the sharded answer is bit-identical to the single-store answer over the
    union — vector (fp32 / fp16, solo and batched, with resident filters), hybrid in one call;
  * the RCCL form at world 1 (`FORCE_RCCL`: ncclCommInitAll + in-place ncclAllGather / ncclAllReduce on one rank) and
    the one-process-per-GPU form (`from_rank`, ncclCommInitRank) run the same code the 8-GPU job runs;
  * the pipelined session (bench.py's timed loop) returns what direct searches return.
"""
import numpy as np
import pytest

import oramacore_amd as oa
import util
from oramacore_amd import _native as N
from oramacore_amd import fulltext as ft
from oramacore_amd.shard_group import FORCE_RCCL, ShardGroup

pytestmark = pytest.mark.gpu


def build_vec_shards(group, rows, doc_ids, cuts, dtype=N.DTYPE_F32):
    stores = []
    for g in range(len(cuts) - 1):
        st = oa.EmbeddingFieldStorage(group.ctx(g), dimensions=rows.shape[1], dtype=dtype)
        st.insert_rows(doc_ids[cuts[g]:cuts[g + 1]], rows[cuts[g]:cuts[g + 1]])
        stores.append(st)
    return stores


@pytest.mark.parametrize("dtype", [N.DTYPE_F32, N.DTYPE_F16], ids=["f32", "f16"])
def test_colocated_vector_shards_equal_single_store(ctx, dtype):
    n, d, k = 6000, 384, 50
    rows = util.gaussian_rows(n, d, seed=3)
    rows[100] = rows[4000]  # equal distances across shards: the tie rule (doc id asc) must survive the merge
    doc_ids = np.arange(n, dtype=np.uint64) * np.uint64(3) + np.uint64(7)
    single = oa.EmbeddingFieldStorage(ctx, dimensions=d, dtype=dtype)
    single.insert_rows(doc_ids, rows)
    qs = util.gaussian_rows(5, d, seed=11)
    cuts = [0, 1500, 1507, 4100, n]
    group = ShardGroup([0] * 4)
    assert group.world == 4 and group.n_local == 4 and not group.uses_rccl
    stores = build_vec_shards(group, rows, doc_ids, cuts, dtype)
    for q in (qs[0], qs):
        ids, dist, cnt = group.vec_search(stores, q, k)
        e_ids, e_dist, e_cnt = single.storage_search(q, k)
        assert np.array_equal(cnt, e_cnt) and np.array_equal(ids, e_ids) and np.array_equal(dist, e_dist)
    # resident filter, one bitmap per shard (each lives with its shard's context)
    bm = oa.AllowBitmap(int(doc_ids.max()) + 1, doc_ids[(np.arange(n) % 3) != 1])
    res = [bm.to_device(group.ctx(i)) for i in range(4)]
    ids, dist, cnt = group.vec_search(stores, qs[1], k, allow=res)
    e_ids, e_dist, e_cnt = single.storage_search(qs[1], k, bm)
    assert np.array_equal(ids, e_ids) and np.array_equal(dist, e_dist) and np.array_equal(cnt, e_cnt)
    # fewer than k rows in total
    ids, dist, cnt = group.vec_search(stores, qs[2], 4096 // 4)
    e_ids, e_dist, e_cnt = single.storage_search(qs[2], 4096 // 4)
    assert np.array_equal(ids, e_ids) and np.array_equal(dist, e_dist)
    for r in res:
        r.close()
    for s in stores:
        s.close()
    single.close()
    group.close()


@pytest.mark.parametrize("mode", ["rccl_world1", "rank_form"])
def test_rccl_path_on_one_rank(ctx, mode):
    """The RCCL calls themselves (in-place all-gather of the candidate blocks, all-reduce of df / min-max) at world 1."""
    if mode == "rccl_world1":
        group = ShardGroup([0], flags=FORCE_RCCL)
    else:
        group = ShardGroup.from_rank(ShardGroup.unique_id(), 0, 1, 0)
    assert group.world == 1 and group.uses_rccl
    n, d, k = 3000, 768, 20
    rows = util.gaussian_rows(n, d, seed=5)
    doc_ids = np.arange(n, dtype=np.uint64)
    st = oa.EmbeddingFieldStorage(group.ctx(0), dimensions=d)
    st.insert_rows(doc_ids, rows)
    qs = util.gaussian_rows(3, d, seed=6)
    ids, dist, cnt = group.vec_search([st], qs, k)
    e_ids, e_dist, e_cnt = st.storage_search(qs, k)
    assert np.array_equal(ids, e_ids) and np.array_equal(dist, e_dist) and np.array_equal(cnt, e_cnt)
    # full-text + hybrid through the all-reduces
    rng = np.random.default_rng(2)
    n_docs, n_tok = 3000, 3
    lens = rng.integers(5, 200, size=n_docs).astype(np.uint32)
    lists = []
    for t in range(n_tok):
        pos = np.sort(rng.choice(n_docs, size=int(rng.integers(200, 1500)), replace=False))
        lists.append(ft.PostingList(field=0, docs=doc_ids[pos], tf=rng.integers(1, 5, size=len(pos)).astype(np.uint32),
                                    field_len=lens[pos]))
    post = ft.PostingsStore(group.ctx(0))
    post.build(doc_ids, [float(lens.mean())], lists)
    refs = [(t, t, 1.0) for t in range(n_tok)]
    got = group.post_search([post], refs, n_tok, float(n_docs), 25)
    exp = post.search(refs, n_tok, float(n_docs), 25)
    assert got[2] == exp[2] and got[0].tolist() == exp[0].tolist() and np.array_equal(got[1].view(np.uint32), exp[1].view(np.uint32))
    got = group.hybrid_search([st], [post], qs[0], 10, 0.0, refs, n_tok, float(n_docs), 25)
    exp = post.hybrid_search(st, qs[0], 10, 0.0, refs, n_tok, float(n_docs), 25)
    assert got[2] == exp[2] and got[0].tolist() == exp[0].tolist() and np.array_equal(got[1].view(np.uint32), exp[1].view(np.uint32))
    group.barrier()
    assert group.allreduce_max(1.25) == 1.25
    # the pipelined session with the exchange forced on (all-gather + K6 on the tail streams)
    sess = group.session([st], qs, 1, k, n_slots=2, force_exchange=True)
    for i in range(6):
        sess.step(i)
    sess.sync()
    for slot, qi in ((0, 4 % 3), (1, 5 % 3)):
        s_ids, s_dist, s_cnt = sess.result(slot)
        assert np.array_equal(s_ids[0], e_ids[qi]) and np.array_equal(s_dist[0], e_dist[qi]) and s_cnt[0] == k
    sess.close()
    post.close()
    st.close()
    group.close()


def test_colocated_hybrid_in_one_call(ctx):
    """orama_shard_hybrid_search over 3 co-located shards == orama_hybrid_search over the union."""
    n, d = 900, 384
    rng = np.random.default_rng(8)
    rows = util.gaussian_rows(n, d, seed=9)
    doc_ids = np.arange(n, dtype=np.uint64)
    q = rows[17] + 0.3 * rows[500]
    lens = rng.integers(5, 100, size=n).astype(np.uint32)
    avg = float(lens.mean())
    n_tok = 2
    pos = [np.sort(rng.choice(n, size=300, replace=False)) for _ in range(n_tok)]
    tfs = [rng.integers(1, 4, size=300).astype(np.uint32) for _ in range(n_tok)]
    single_vec = oa.EmbeddingFieldStorage(ctx, dimensions=d)
    single_vec.insert_rows(doc_ids, rows)
    single_post = ft.PostingsStore(ctx)
    single_post.build(doc_ids, [avg], [ft.PostingList(field=0, docs=doc_ids[p], tf=t, field_len=lens[p])
                                       for p, t in zip(pos, tfs)])
    cuts = [0, 333, 600, n]
    group = ShardGroup([0, 0, 0])
    vecs = build_vec_shards(group, rows, doc_ids, cuts)
    posts = []
    for g in range(3):
        lo, hi = cuts[g], cuts[g + 1]
        lists = []
        for p, t in zip(pos, tfs):
            m = (p >= lo) & (p < hi)
            lists.append(ft.PostingList(field=0, docs=doc_ids[p[m]], tf=t[m], field_len=lens[p[m]]))
        ps = ft.PostingsStore(group.ctx(g))
        ps.build(doc_ids[lo:hi], [avg], lists)  # index-wide average length
        posts.append(ps)
    refs = [(t, t, 1.0) for t in range(n_tok)]
    for limit, sim in ((10, 0.0), (40, 0.05)):
        got = group.hybrid_search(vecs, posts, q, limit, sim, refs, n_tok, float(n), 30)
        exp = single_post.hybrid_search(single_vec, q, limit, sim, refs, n_tok, float(n), 30)
        assert got[2] == exp[2] and got[0].tolist() == exp[0].tolist()
        assert np.array_equal(got[1].view(np.uint32), exp[1].view(np.uint32))
    # session over co-located shards: the gathered buffer is shared, K6 runs once
    qs = util.gaussian_rows(4, d, seed=21)
    sess = group.session(vecs, qs, 2, 15, n_slots=2)
    for i in range(4):
        sess.step(i)
    sess.sync()
    e_ids, e_dist, _ = single_vec.storage_search(qs, 15)
    for slot, step in ((0, 2), (1, 3)):
        s_ids, s_dist, s_cnt = sess.result(slot)
        sl = slice((step % 2) * 2, (step % 2) * 2 + 2)
        assert np.array_equal(s_ids, e_ids[sl]) and np.array_equal(s_dist, e_dist[sl])
    sess.close()
    for s in vecs + posts + [single_vec, single_post]:
        s.close()
    group.close()


def test_concurrent_callers_on_a_colocated_group(ctx):
    """Many threads at once on ONE 4-shard group (the re-entrancy `search(&self)` assumes, read/collection.rs:846-884):
    vector searches of 1..20 queries, full-text, hybrid, requests through the group's batcher and a pipelined session —
    every answer bit-identical to the single store's, no deadlock; the group really ran calls side by side (more than
    one lane in use)."""
    import threading

    n, d, n_tok = 4000, 128, 3
    rng = np.random.default_rng(18)
    rows = util.gaussian_rows(n, d, seed=19)
    doc_ids = np.arange(n, dtype=np.uint64) * 2 + 7
    lens = rng.integers(5, 100, size=n).astype(np.uint32)
    avg = float(lens.mean())
    pos = [np.sort(rng.choice(n, size=sz, replace=False)) for sz in (1500, 900, 400)]
    tfs = [rng.integers(1, 4, size=len(p)).astype(np.uint32) for p in pos]
    single_vec = oa.EmbeddingFieldStorage(ctx, dimensions=d)
    single_vec.insert_rows(doc_ids, rows)
    single_post = ft.PostingsStore(ctx)
    single_post.build(doc_ids, [avg], [ft.PostingList(field=0, docs=doc_ids[p], tf=t, field_len=lens[p]) for p, t in zip(pos, tfs)])
    cuts = [0, 1000, 2100, 3000, n]
    group = ShardGroup([0, 0, 0, 0])
    vecs = build_vec_shards(group, rows, doc_ids, cuts)
    posts = []
    for g in range(4):
        lo, hi = cuts[g], cuts[g + 1]
        lists = []
        for p, t in zip(pos, tfs):
            m = (p >= lo) & (p < hi)
            lists.append(ft.PostingList(field=0, docs=doc_ids[p[m]], tf=t[m], field_len=lens[p[m]]))
        ps = ft.PostingsStore(group.ctx(g))
        ps.build(doc_ids[lo:hi], [avg], lists)
        posts.append(ps)
    refs = [(t, t, 1.0) for t in range(n_tok)]
    queries = util.gaussian_rows(64, d, seed=23)
    # expected answers from the single stores
    e_vec = {nq: single_vec.storage_search(queries[:nq], 10) for nq in (1, 5, 20)}
    e_one = [single_vec.storage_search(queries[i], 7) for i in range(16)]
    e_post = single_post.search(refs, n_tok, float(n), 25)
    e_hyb = [single_post.hybrid_search(single_vec, queries[i], 12, 0.0, refs, n_tok, float(n), 20) for i in range(8)]
    batcher = group.batcher(vecs, max_batch=32)
    errors, done = [], {"vec": 0, "post": 0, "hyb": 0, "bat": 0}
    lock = threading.Lock()
    deadline = [None]

    def same(a, b):
        return np.array_equal(np.asarray(a[0]), np.asarray(b[0])) and np.array_equal(np.asarray(a[1]).view(np.uint32), np.asarray(b[1]).view(np.uint32))

    def worker(kind, seed):
        import time
        r = np.random.default_rng(seed)
        try:
            while time.monotonic() < deadline[0] and not errors:
                if kind == "vec":
                    nq = int(r.choice([1, 5, 20]))
                    got = group.vec_search(vecs, queries[:nq], 10)
                    assert same(got, e_vec[nq]) and np.array_equal(got[2], e_vec[nq][2])
                elif kind == "post":
                    got = group.post_search(posts, refs, n_tok, float(n), 25)
                    assert got[2] == e_post[2] and same(got, e_post)
                elif kind == "hyb":
                    i = int(r.integers(8))
                    got = group.hybrid_search(vecs, posts, queries[i], 12, 0.0, refs, n_tok, float(n), 20)
                    assert got[2] == e_hyb[i][2] and same(got, e_hyb[i])
                else:
                    i = int(r.integers(16))
                    ids, dist = batcher.search(queries[i], 7)
                    assert np.array_equal(ids, e_one[i][0][0]) and np.array_equal(dist.view(np.uint32), e_one[i][1][0].view(np.uint32))
                with lock:
                    done[kind] += 1
        except Exception as e:  # noqa: BLE001
            errors.append((kind, repr(e)))

    import time
    deadline[0] = time.monotonic() + 4.0
    threads = [threading.Thread(target=worker, args=(kind, 100 + i)) for i, kind in
               enumerate(["vec"] * 4 + ["post"] * 3 + ["hyb"] * 3 + ["bat"] * 6)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    assert not any(t.is_alive() for t in threads), "a caller is stuck"
    assert not errors, errors[:3]
    assert all(v > 0 for v in done.values()), done
    assert batcher.stats()["requests"] == done["bat"]
    lanes = group.lanes()
    assert lanes["max"] >= 2 and lanes["created"] >= 2, lanes
    batcher.close()
    for s in posts + vecs:
        s.close()
    single_post.close()
    single_vec.close()
    group.close()


def test_shadow_store_shards_take_the_two_stage_plan(ctx):
    """Shards that keep an fp16 shadow answer a sharded search with the two-stage exact plan (begun on every shard, then
    joined): same (ids, distance bits, counts) as ONE plain fp32 store over the union — 1, 5 and 70 queries, a filter, near
    duplicates that force the proof to fail (the plain scan re-answers those queries) — and the plan really ran."""
    n, d, k = 9000, 128, 40
    rows = util.gaussian_rows(n, d, seed=31)
    rows[2000:2300] = rows[100] + np.float32(1e-4) * util.gaussian_rows(300, d, seed=32)  # 300 near-duplicates of row 100
    doc_ids = np.arange(n, dtype=np.uint64) * 2 + 1
    single = oa.EmbeddingFieldStorage(ctx, dimensions=d)
    single.insert_rows(doc_ids, rows)
    cuts = [0, 2500, 5200, n]
    group = ShardGroup([0, 0, 0])
    for g in range(3):
        group.ctx(g).set_two_stage(True, always=True)
    shards = build_vec_shards(group, rows, doc_ids, cuts, dtype=N.DTYPE_F32_SHADOW16)
    qs = util.gaussian_rows(70, d, seed=33)
    qs[3] = rows[100]  # its top-40 are the near-duplicates: more candidates inside the error band than the list holds
    for nq in (1, 5, 70):
        got = group.vec_search(shards, qs[:nq], k)
        exp = single.storage_search(qs[:nq], k)
        assert np.array_equal(got[2], exp[2]) and np.array_equal(got[0], exp[0]), nq
        assert np.array_equal(got[1].view(np.uint32), exp[1].view(np.uint32)), nq
    keep = (np.arange(n) % 3) != 0
    bm_single = oa.AllowBitmap(int(doc_ids.max()) + 1, doc_ids[keep])
    toks = [oa.AllowBitmap(int(doc_ids.max()) + 1, doc_ids[keep]).to_device(group.ctx(g)) for g in range(3)]
    got = group.vec_search(shards, qs[:5], k, allow=toks)
    exp = single.storage_search(qs[:5], k, bm_single)
    assert np.array_equal(got[0], exp[0]) and np.array_equal(got[1].view(np.uint32), exp[1].view(np.uint32))
    infos = [s.info() for s in shards]
    assert all(i["two_stage_queries"] >= 81 for i in infos), infos
    assert sum(i["two_stage_fallbacks"] for i in infos) >= 1, infos  # the near-duplicate query on the shard that owns them
    for t in toks:
        t.close()
    for s in shards:
        s.close()
    single.close()
    group.close()
