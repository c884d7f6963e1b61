"""One index sharded over G doc-id ranges, full-text and hybrid, through orama_shard_post_search: the
exchange (df all-reduce SUM, min/max all-reduce MAX, block all-gather + K6) runs INSIDE the library.

The G shard stores live on the one GPU of the test box (a co-located shard group: same stages, same buffers, the
reductions are device-local kernels instead of RCCL calls; tests/test_shard_group_gpu.py runs the RCCL form at
world 1).  Bar: bit-identical to the oracle over the union (and therefore to the single-store search): ids, scores
and match count.
"""
import numpy as np
import pytest

import oramacore_amd as oa
import util
from oracle import oracle as orc
from oramacore_amd import fulltext as ft
from oramacore_amd.shard_group import ShardGroup
from test_fulltext_gpu import bits, oracle_topk
from test_oracle_golden import bm25_synth_entries

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def synth():
    meta = util.load_json("bm25_synth.json")
    fields = util.mg.zipf_corpus(meta["n_docs"], meta["vocab"], meta["n_fields"], seed=meta["seed"])
    doc_ids = np.arange(meta["n_docs"], dtype=np.uint64) * np.uint64(meta["doc_id_mul"]) + np.uint64(meta["doc_id_add"])
    allow = (util.hash_u64(doc_ids + np.uint64(5)) % np.uint64(3)) != 0
    return meta, fields, doc_ids, allow


def build_shards(ctx, meta, fields, doc_ids, cuts):
    """Shard g holds documents [cuts[g], cuts[g+1]) (positions in doc_ids); every shard has every list id."""
    keys = [(f, term) for f in range(meta["n_fields"]) for term in sorted(fields[f]["postings"])]
    list_id = {key: i for i, key in enumerate(keys)}
    avg = [fields[f]["avg"] for f in range(meta["n_fields"])]  # index-wide averages
    stores = []
    for g in range(len(cuts) - 1):
        lo, hi = cuts[g], cuts[g + 1]
        lists = []
        for f, term in keys:
            pl = [p for p in fields[f]["postings"][term] if lo <= p[0] < hi]
            dix = np.array([p[0] for p in pl], dtype=np.int64)
            lists.append(ft.PostingList(field=f, docs=doc_ids[dix], tf=np.array([p[1] for p in pl], dtype=np.uint32),
                                        field_len=fields[f]["lens"][dix] if len(dix) else np.zeros(0, np.uint32)))
        st = ft.PostingsStore(ctx)
        st.build(doc_ids[lo:hi], avg, lists)
        stores.append(st)
    return stores, list_id


def run_sharded(group, stores, refs, n_tok, total_docs, top_k, threshold=None, allow=None, vector=None,
                apply_omc=True):
    """One sharded query through the library (the caller of round 1's staged entry points, now one C call)."""
    res_allow = None
    if allow is not None:
        res_allow = [allow.to_device(group.ctx(i)) for i in range(group.n_local)]
    try:
        return group.post_search(stores, refs, n_tok, total_docs, top_k, threshold=threshold, allow=res_allow,
                                 apply_omc=apply_omc, vector=vector)
    finally:
        for a in res_allow or []:
            a.close()


def refs_of(meta, list_id, case):
    return [(ti, list_id[(f, t)], meta["boosts"][f]) for ti, t in enumerate(case["terms"])
            for f in range(meta["n_fields"]) if (f, t) in list_id]


@pytest.mark.parametrize("cuts_frac", [(0.0, 0.5, 1.0), (0.0, 0.13, 0.5, 0.51, 1.0)], ids=["2shards", "4ragged"])
def test_sharded_bm25_bit_exact(ctx, synth, cuts_frac):
    meta, fields, doc_ids, allow = synth
    cuts = [int(round(f * meta["n_docs"])) for f in cuts_frac]
    group = ShardGroup([0] * (len(cuts) - 1))
    stores, list_id = build_shards(group.ctx(0), meta, fields, doc_ids, cuts)
    bm = oa.AllowBitmap(int(doc_ids.max()) + 1, doc_ids[allow])
    for case in meta["cases"][::3]:
        refs = refs_of(meta, list_id, case)
        entries = bm25_synth_entries(meta, fields, case, doc_ids, allow)
        n_tok = len(case["terms"])
        ids, sc, count = run_sharded(group, stores, refs, n_tok, float(meta["n_docs"]), 50, case["threshold"],
                                     allow=bm if case["filter"] else None)
        od, os_, ocount = oracle_topk(entries, n_tok, meta["n_docs"], 50, case["threshold"])
        assert count == ocount, (case["query"], case["threshold"], case["filter"])
        assert ids.tolist() == od.tolist(), (case["query"], case["threshold"], case["filter"])
        assert np.array_equal(bits(sc), bits(os_))
    for s in stores:
        s.close()
    group.close()


def test_sharded_hybrid_and_omc_bit_exact(ctx, synth):
    meta, fields, doc_ids, allow = synth
    cuts = [0, meta["n_docs"] // 3, meta["n_docs"] // 3 + 7, meta["n_docs"]]
    group = ShardGroup([0] * (len(cuts) - 1))
    stores, list_id = build_shards(group.ctx(0), meta, fields, doc_ids, cuts)
    rng = np.random.default_rng(11)
    for ci in (0, 12, 24, 30):
        case = {**meta["cases"][ci], "filter": False}
        refs = refs_of(meta, list_id, case)
        entries = bm25_synth_entries(meta, fields, case, doc_ids, allow)
        n_tok = len(case["terms"])
        fd, fs = orc.search_full_text(entries, n_tok, float(meta["n_docs"]), 1.2, case["threshold"])
        inside = rng.choice(fd, size=min(6, len(fd)), replace=False) if len(fd) else np.array([], dtype=np.uint64)
        outside = np.setdiff1d(doc_ids[::97], fd)[:6]  # vector-only docs spread over all shards
        vec = {int(d): float(s) for d, s in zip(np.concatenate([inside, outside]),
                                                 rng.uniform(-0.2, 1.0, size=len(inside) + len(outside)))}
        for omc in (None, {int(doc_ids[3]): 2.0, int(doc_ids[-2]): 4.0, int(fd[0]) if len(fd) else 1: 0.5}):
            od, os_ = orc.normalize_and_combine(list(vec), list(vec.values()), fd, fs)
            if omc:
                os_ = orc.apply_omc(od, os_, list(omc), list(omc.values()))
                for s in stores:
                    s.set_omc(omc)
            td, ts = orc.top_n(od, os_, 30)
            ids, sc, count = run_sharded(group, stores, refs, n_tok, float(meta["n_docs"]), 30, case["threshold"],
                                         vector=vec, apply_omc=omc is not None)
            assert count == len(od), (ci, omc)
            assert ids.tolist() == td.tolist(), (ci, omc)
            assert np.array_equal(bits(sc), bits(ts))
    # OMC without hybrid
    case = meta["cases"][12]
    refs = refs_of(meta, list_id, case)
    entries = bm25_synth_entries(meta, fields, {**case, "filter": False}, doc_ids, allow)
    omc = {int(doc_ids[5]): 10.0, int(doc_ids[1500]): 0.25}
    for s in stores:
        s.set_omc(omc)
    ids, sc, count = run_sharded(group, stores, refs, len(case["terms"]), float(meta["n_docs"]), 100)
    od, os_, ocount = oracle_topk(entries, len(case["terms"]), meta["n_docs"], 100, omc=omc)
    assert count == ocount and ids.tolist() == od.tolist() and np.array_equal(bits(sc), bits(os_))
    for s in stores:
        s.close()
    group.close()


def test_single_shard_group_equals_store(ctx, synth):
    """A group of ONE shard: the same stages (begin / score / finish / merge) — equals PostingsStore.search."""
    meta, fields, doc_ids, allow = synth
    group = ShardGroup([0])
    stores, list_id = build_shards(group.ctx(0), meta, fields, doc_ids, [0, meta["n_docs"]])
    case = meta["cases"][12]
    refs = refs_of(meta, list_id, case)
    n_tok = len(case["terms"])
    ids, sc, count = group.post_search(stores, refs, n_tok, float(meta["n_docs"]), 40)
    ids1, sc1, count1 = stores[0].search(refs, n_tok, float(meta["n_docs"]), 40)
    assert count == count1 and ids.tolist() == ids1.tolist() and np.array_equal(bits(sc), bits(sc1))
    vec = {int(doc_ids[1]): 0.9, int(ids1[0]): 0.4}
    ids, sc, count = group.post_search(stores, refs, n_tok, float(meta["n_docs"]), 40, vector=vec)
    ids1, sc1, count1 = stores[0].search(refs, n_tok, float(meta["n_docs"]), 40, vector=vec)
    assert count == count1 and ids.tolist() == ids1.tolist() and np.array_equal(bits(sc), bits(sc1))
    stores[0].close()
    group.close()


def test_staged_query_order_is_enforced(ctx, synth):
    meta, fields, doc_ids, allow = synth
    stores, list_id = build_shards(ctx, meta, fields, doc_ids, [0, meta["n_docs"]])
    lib = oa._native.load()
    d_df = oa.DeviceBuffer(ctx, 64 * 4)
    block = oa.DeviceBuffer(ctx, int(lib.orama_post_block_bytes(5)))
    stream = oa.Stream(ctx)
    q = stores[0].staged_query([(0, 0, 1.0)], 1, 100.0, 5, d_df.ptr, stream.ptr)
    with pytest.raises(oa.OramaError):
        q.finish(None, None, block.ptr)  # score not called yet
    q.score(np.array([3], dtype=np.uint32), None)
    with pytest.raises(oa.OramaError):
        q.score(np.array([3], dtype=np.uint32), None)
    q.finish(None, None, block.ptr)
    q.end()
    stream.close()
    stores[0].close()


def test_sharded_fulltext_batches_equal_the_single_store(ctx):
    """orama_shard_post_search_batch over 3 ragged co-located shards == orama_post_search_batch over the union, bit for bit:
    queries the range scorer takes on every shard with the index-wide df summed on the host (one list per token, no filter;
    70 of them: three sets of launches per shard, shards side by side), and queries that go one by one through the staged
    sharded query (two lists for a token, a filter); thresholds, OMC, k from 1 to 150; a malformed query fails alone.
    The group's full-text request batcher answers the same from concurrent callers."""
    import threading

    from oramacore_amd.shard_group import ShardGroup

    rng = np.random.default_rng(41)
    n, n_lists = 30_000, 10
    doc_ids = np.arange(n, dtype=np.uint64) * 3 + 2
    lens = rng.integers(5, 300, size=n).astype(np.uint32)
    avg = float(lens.mean())
    pos = [np.sort(rng.choice(n, size=int(sz), replace=False)) for sz in rng.integers(200, 9000, size=n_lists)]
    tfs = [rng.integers(1, 6, size=len(p)).astype(np.uint32) for p in pos]
    single = ft.PostingsStore(ctx)
    single.build(doc_ids, [avg], [ft.PostingList(field=0, docs=doc_ids[p], tf=t, field_len=lens[p]) for p, t in zip(pos, tfs)])
    omc = {int(doc_ids[i]): float(np.float32(m)) for i, m in zip(rng.choice(n, size=40, replace=False), rng.uniform(0.2, 30.0, size=40))}
    single.set_omc(omc)
    cuts = [0, 7000, 19000, n]
    group = ShardGroup([0, 0, 0])
    shards = []
    for g in range(3):
        lo, hi = cuts[g], cuts[g + 1]
        lists = []
        for p, t in zip(pos, tfs):
            m = (p >= lo) & (p < hi)
            lists.append(ft.PostingList(field=0, docs=doc_ids[p[m]], tf=t[m], field_len=lens[p[m]]))
        ps = ft.PostingsStore(group.ctx(g))
        ps.build(doc_ids[lo:hi], [avg], lists)  # index-wide average length
        ps.set_omc({d: m for d, m in omc.items() if doc_ids[lo] <= d <= doc_ids[hi - 1]})
        shards.append(ps)
    queries = []
    for i in range(70):
        n_tok = 1 + i % 4
        ls = rng.choice(n_lists, size=n_tok, replace=False)
        refs = [(t, int(l), float(np.float32(1.0 + 0.5 * (i % 3)))) for t, l in enumerate(ls)]
        queries.append((refs, n_tok, (1 if n_tok > 1 and i % 5 == 0 else None), [1, 10, 50, 150][i % 4]))
    queries.append(([(0, 1, 1.0), (0, 2, 1.0), (1, 3, 1.0)], 2, None, 25))  # two lists for token 0: one by one
    queries.append(([(0, 4, 1.0)], 1, None, 10))
    for use_omc in (False, True):
        exp = single.search_batch(queries, float(n), 150, apply_omc=use_omc)
        got = group.post_search_batch(shards, queries, float(n), 150, apply_omc=use_omc)
        for i, ((gi, gs, gc), (ei, es, ec)) in enumerate(zip(got, exp)):
            assert gc == ec and gi.tolist() == ei.tolist(), (use_omc, i)
            assert np.array_equal(gs.view(np.uint32), es.view(np.uint32)), (use_omc, i)
    # the eligible queries really ran on the range scorer, without a per-record launch
    gctx = group.ctx(0)
    gctx.prof_reset()
    gctx.prof_enable(True)
    group.post_search_batch(shards, queries[:70], float(n), 150)
    gctx.prof_enable(False)
    assert gctx.prof_get("bm25_range_score")[1] >= 9 and gctx.prof_get("bm25_accumulate")[1] == 0  # 3 shards x 3 sets of launches
    # a filter: every query goes through the staged sharded query
    keep = (np.arange(n) % 4) != 1
    bm = oa.AllowBitmap(int(doc_ids.max()) + 1, doc_ids[keep])
    toks = [oa.AllowBitmap(int(doc_ids.max()) + 1, doc_ids[keep]).to_device(group.ctx(g)) for g in range(3)]
    exp = single.search_batch(queries[:6], float(n), 150, allow=bm)
    got = group.post_search_batch(shards, queries[:6], float(n), 150, allow=toks)
    for (gi, gs, gc), (ei, es, ec) in zip(got, exp):
        assert gc == ec and gi.tolist() == ei.tolist() and np.array_equal(gs.view(np.uint32), es.view(np.uint32))
    # a malformed query (token index beyond n_tokens) fails alone
    bad = queries[:3] + [([(5, 0, 1.0)], 2, None, 5)] + queries[3:5]
    res, st = group.post_search_batch(shards, bad, float(n), 150, statuses=True)
    assert st.tolist() == [0, 0, 0, oa._native.ORAMA_ERR_INVALID, 0, 0] and len(res[3][0]) == 0
    exp = single.search_batch(queries[:5], float(n), 150)
    for (gi, gs, gc), (ei, es, ec) in zip(res[:3] + res[4:], exp):
        assert gc == ec and gi.tolist() == ei.tolist()
    # the request batcher in front of the group, concurrent callers
    batcher = group.post_batcher(shards, max_batch=64)
    exp = single.search_batch(queries[:40], float(n), 150)
    errors = []

    def caller(i):
        try:
            refs, n_tok, thr, k = queries[i]
            ids, sc, cnt = batcher.search(refs, n_tok, float(n), k, thr)
            assert cnt == exp[i][2] and ids.tolist() == exp[i][0].tolist() and np.array_equal(sc.view(np.uint32), exp[i][1].view(np.uint32))
        except Exception as e:  # noqa: BLE001
            errors.append((i, repr(e)))

    threads = [threading.Thread(target=caller, args=(i,)) for i in range(40)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    assert not errors, errors[:3]
    assert batcher.stats()["requests"] == 40
    batcher.close()
    for t in toks:
        t.close()
    for s in shards:
        s.close()
    single.close()
    group.close()
