"""GPU parity of the BM25F path (K3 + K4) and the hybrid combine (K5) through the C ABI.

Bar (north_star): bit-exact docID ordering and scores for pure BM25 — the HIP path must reproduce the
oracle's f32 arithmetic exactly (same operation order, libm idf), not approximately.
"""
import numpy as np
import pytest

import oramacore_amd as oa
import util
from oracle import oracle as orc
from oramacore_amd import fulltext as ft
from test_oracle_golden import bm25_synth_entries

pytestmark = pytest.mark.gpu

F = np.float32


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def oracle_topk(entries, n_tokens, n_docs, top_k, threshold=None, omc=None):
    docs, scores = orc.search_full_text(entries, n_tokens, float(n_docs), 1.2, threshold)
    if omc:
        scores = orc.apply_omc(docs, scores, list(omc), list(omc.values()))
    td, ts = orc.top_n(docs, scores, top_k)
    return td, ts, len(docs)


# ----------------------------------------------------------------------------- reference known answers
def test_reference_kat_two_fields_through_scorer_mirror(ctx):
    """bm25.rs:911-983 (test_canonical_bm25f_single_term_two_fields) driven through the BM25Scorer mirror:
    expected value from the reference test body, tolerance 1e-5 as in the reference; and bit-equal to the oracle."""
    case = [c for c in util.load_json("bm25_kat.json")["cases"]
            if c["name"] == "test_canonical_bm25f_single_term_two_fields"][0]
    f0, f1 = case["fields"]
    scorer = ft.BM25Scorer.plain(ctx)
    scorer.reset_term()
    # the in-tree caller derives df from the postings (token_score.rs:262-275): 10 docs hold the term
    for d in range(1, 11):
        scorer.add_precomputed_field(d, orc.bm25f_normalized_tf(f0["tf"], f0["len"], f0["avglen"], f0["b"]), f0["weight"])
    scorer.add_precomputed_field(1, orc.bm25f_normalized_tf(f1["tf"], f1["len"], f1["avglen"], f1["b"]), f1["weight"])
    scorer.finalize_term_plain(10, 100.0, 1.2, 1.0)
    scorer.next_term()
    scores = scorer.get_scores()
    assert len(scores) == 10
    assert abs(float(scores[1]) - case["expect"]["1"]) <= case["tol"]
    n0 = F(F(f0["weight"]) * orc.bm25f_normalized_tf(f0["tf"], f0["len"], f0["avglen"], f0["b"]))
    n1 = F(F(f1["weight"]) * orc.bm25f_normalized_tf(f1["tf"], f1["len"], f1["avglen"], f1["b"]))
    od, os_ = orc.search_full_text([(0, list(range(1, 11)), [n0] * 10), (0, [1], [n1])], 1, 100.0)
    assert {int(d): int(bits([s])[0]) for d, s in zip(od, os_)} == {d: int(bits([s])[0]) for d, s in scores.items()}


def test_reference_kat_basic_legacy_formula(ctx):
    """bm25.rs:533-563: single field, tf=5, len=avglen=100, N=100, df=10 → idf*(k+1)*5/(k+5) within 1e-6."""
    case = util.load_json("bm25_kat.json")["cases"][0]
    a = case["adds"][0]
    ntf = orc.bm25f_normalized_tf(a["tf"], a["len"], a["avglen"], a["b"])
    ids, sc, count = ft.bm25_score(ctx, [(0, list(range(1, 11)), [ntf] * 10)], 1, a["total_docs"], 10)
    assert count == 10 and ids.tolist() == list(range(1, 11))  # equal scores → doc id ascending
    assert abs(float(sc[0]) - case["expect"]["1"]) <= case["tol"]
    assert len(set(bits(sc).tolist())) == 1


# ----------------------------------------------------------------------------- synthetic corpus, both seams
@pytest.fixture(scope="module")
def synth():
    meta = util.load_json("bm25_synth.json")
    fields = util.mg.zipf_corpus(meta["n_docs"], meta["vocab"], meta["n_fields"], seed=meta["seed"])
    doc_ids = np.arange(meta["n_docs"], dtype=np.uint64) * np.uint64(meta["doc_id_mul"]) + np.uint64(meta["doc_id_add"])
    allow = (util.hash_u64(doc_ids + np.uint64(5)) % np.uint64(3)) != 0
    return meta, fields, doc_ids, allow


def build_store(ctx, meta, fields, doc_ids):
    """Resident postings: list id = field * vocab + term."""
    lists = []
    list_id = {}
    for f in range(meta["n_fields"]):
        for term in sorted(fields[f]["postings"]):
            pl = fields[f]["postings"][term]
            dix = np.array([p[0] for p in pl], dtype=np.int64)
            list_id[(f, term)] = len(lists)
            lists.append(ft.PostingList(field=f, docs=doc_ids[dix], tf=np.array([p[1] for p in pl]),
                                        field_len=fields[f]["lens"][dix]))
    store = ft.PostingsStore(ctx)
    store.build(doc_ids, [fields[f]["avg"] for f in range(meta["n_fields"])], lists)
    return store, list_id


def test_bm25_synth_seam_i_bit_exact(ctx, synth):
    meta, fields, doc_ids, allow = synth
    for case in meta["cases"]:
        entries = bm25_synth_entries(meta, fields, case, doc_ids, allow)
        n_tok = len(case["terms"])
        for top_k in (1, 20, 500):
            ids, sc, count = ft.bm25_score(ctx, entries, n_tok, float(meta["n_docs"]), top_k, case["threshold"])
            od, os_, ocount = oracle_topk(entries, n_tok, meta["n_docs"], top_k, case["threshold"])
            assert count == ocount == case["count"]
            assert ids.tolist() == od.tolist(), (case["query"], case["threshold"], case["filter"], top_k)
            assert np.array_equal(bits(sc), bits(os_))
        if case["threshold"] is None and not case["filter"]:
            assert ids[:20].tolist() == od[:20].tolist()


def test_bm25_synth_resident_bit_exact(ctx, synth):
    """Seam (ii): ntf computed on the device from (tf, len) + field boost, filter applied to postings, df counted
    on the device, idf from the host-built table — still bit-identical to the oracle fed with host ntf."""
    meta, fields, doc_ids, allow = synth
    store, list_id = build_store(ctx, meta, fields, doc_ids)
    bm = oa.AllowBitmap(int(doc_ids.max()) + 1, doc_ids[allow])
    for case in meta["cases"]:
        refs = []
        for ti, term in enumerate(case["terms"]):
            for f in range(meta["n_fields"]):
                if (f, term) in list_id:
                    refs.append((ti, list_id[(f, term)], meta["boosts"][f]))
        entries = bm25_synth_entries(meta, fields, case, doc_ids, allow)
        n_tok = len(case["terms"])
        ids, sc, count = store.search(refs, n_tok, float(meta["n_docs"]), 50, case["threshold"],
                                      allow=bm if case["filter"] else None)
        od, os_, ocount = oracle_topk(entries, n_tok, meta["n_docs"], 50, case["threshold"])
        assert count == ocount
        assert ids.tolist() == od.tolist(), (case["query"], case["threshold"], case["filter"])
        assert np.array_equal(bits(sc), bits(os_))
    # golden cross-check (independent numpy restatement): count + top ids where scores are isolated
    case = meta["cases"][0]
    refs = [(ti, list_id[(f, t)], meta["boosts"][f]) for ti, t in enumerate(case["terms"])
            for f in range(meta["n_fields"]) if (f, t) in list_id]
    ids, sc, count = store.search(refs, len(case["terms"]), float(meta["n_docs"]), 20, None)
    assert count == case["count"]
    assert np.allclose(sc, np.array(case["top_scores"], dtype=np.float32), rtol=2e-6, atol=1e-6)
    store.close()


def test_omc_multipliers(ctx, synth):
    """apply_omc_multipliers (search.rs:39-48): ratios 0.25/0.5/2/5/10 as in src/tests/omc_test.rs:485-553."""
    meta, fields, doc_ids, allow = synth
    case = meta["cases"][12]  # 3-token query
    entries = bm25_synth_entries(meta, fields, {**case, "filter": False}, doc_ids, allow)
    n_tok = len(case["terms"])
    base_ids, base_sc, count = ft.bm25_score(ctx, entries, n_tok, float(meta["n_docs"]), 2000)
    mult = [0.25, 0.5, 2.0, 5.0, 10.0]
    omc = {int(base_ids[i]): m for i, m in enumerate(mult)}
    omc[10**12] = 3.0  # multiplier of a doc that is not in the result: ignored
    ids, sc, count2 = ft.bm25_score(ctx, entries, n_tok, float(meta["n_docs"]), 2000, omc=omc)
    od, os_, _ = oracle_topk(entries, n_tok, meta["n_docs"], 2000, omc=omc)
    assert count2 == count and ids.tolist() == od.tolist() and np.array_equal(bits(sc), bits(os_))
    got = dict(zip(ids.tolist(), sc.tolist()))
    for i, m in enumerate(mult):
        assert abs(got[int(base_ids[i])] / float(base_sc[i]) - m) < 1e-3
    # resident store: dense OMC
    store, list_id = build_store(ctx, meta, fields, doc_ids)
    store.set_omc(omc)
    refs = [(ti, list_id[(f, t)], meta["boosts"][f]) for ti, t in enumerate(case["terms"])
            for f in range(meta["n_fields"]) if (f, t) in list_id]
    ids2, sc2, _ = store.search(refs, n_tok, float(meta["n_docs"]), 2000)
    assert ids2.tolist() == od.tolist() and np.array_equal(bits(sc2), bits(os_))
    ids3, sc3, _ = store.search(refs, n_tok, float(meta["n_docs"]), 2000, apply_omc=False)
    assert ids3.tolist() == base_ids.tolist()
    store.close()


def test_fulltext_ordinal_replica(ctx):
    """src/tests/fulltext_search.rs:192-251 through the resident store: 100 docs, doc i holds "text " x (i+1);
    query "text" → ranking 99, 98, 97 …, limit 10, count 100."""
    g = util.load_json("fulltext_ordinal.json")
    n = g["n"]
    docs = np.arange(n, dtype=np.uint64)
    lens = np.arange(1, n + 1)
    store = ft.PostingsStore(ctx)
    store.build(docs, [g["avg"]], [ft.PostingList(field=0, docs=docs, tf=lens, field_len=lens)])
    ids, sc, count = store.search([(0, 0, 1.0)], 1, float(n), 10)
    assert count == 100 and ids.tolist() == list(range(99, 89, -1)) == g["top_ids"]
    assert np.allclose(sc, g["top_scores"], rtol=1e-6)
    # offset paging (fulltext_search.rs:254-335): top (limit+offset) then skip(offset).take(limit)
    ids2, _, _ = store.search([(0, 0, 1.0)], 1, float(n), 25)
    assert ids2[20:25].tolist() == [79, 78, 77, 76, 75]
    store.close()


def test_all_scores_equal_large_tie(ctx):
    """benches/fulltext_simple.rs shape: every doc holds the query term once with the same length → all scores
    equal; the top-10 must be the 10 lowest doc ids (declared tie rule) and count = N."""
    n = 200_000
    docs = np.arange(n, dtype=np.uint64) + np.uint64(1)
    ntf = np.full(n, orc.bm25f_normalized_tf(1, 7, 7.0, 0.75), dtype=np.float32)
    ids, sc, count = ft.bm25_score(ctx, [(0, docs, ntf)], 1, float(n), 10)
    assert count == n and ids.tolist() == list(range(1, 11)) and len(set(bits(sc).tolist())) == 1
    od, os_, _ = oracle_topk([(0, docs, ntf)], 1, n, 10)
    assert np.array_equal(bits(sc), bits(os_))


def test_edge_cases(ctx):
    # token without postings contributes nothing; empty query → empty result
    ids, sc, count = ft.bm25_score(ctx, [(0, [], [])], 1, 100.0, 10)
    assert count == 0 and len(ids) == 0
    # subnormal / zero / negative-zero S are skipped (is_normal), NaN term scores are skipped
    docs = [1, 2, 3, 4, 5]
    ntf = np.array([1e-45, 0.0, -0.0, 1.0, np.inf], dtype=np.float32)
    ids, sc, count = ft.bm25_score(ctx, [(0, docs, ntf)], 1, 100.0, 10)
    od, os_, ocount = oracle_topk([(0, docs, ntf)], 1, 100, 10)
    assert count == ocount and ids.tolist() == od.tolist() and np.array_equal(bits(sc), bits(os_))
    # top_k = 0 still returns the match count
    ids, sc, count = ft.bm25_score(ctx, [(0, [1, 2, 3], [1.0, 2.0, 3.0])], 1, 100.0, 0)
    assert count == 3 and len(ids) == 0
    # non-dense 64-bit doc ids take the sorted-set path
    big = np.array([5, 2**40, 2**41 + 7, 2**63 + 11], dtype=np.uint64)
    ids, sc, count = ft.bm25_score(ctx, [(0, big, [1.0, 3.0, 2.0, 3.0]), (1, big[1:3], [0.5, 0.25])], 2, 1000.0, 4)
    od, os_, ocount = oracle_topk([(0, big, [1.0, 3.0, 2.0, 3.0]), (1, big[1:3], [0.5, 0.25])], 2, 1000, 4)
    assert count == ocount == 4 and ids.tolist() == od.tolist() and np.array_equal(bits(sc), bits(os_))


def test_many_tokens_and_three_entries_per_token(ctx):
    """40 tokens (mask wraps at 32 like Rust's `1 << term_index` on u32 in release builds) and 3 entries per token
    hitting the same docs: the f32 accumulation order (entry order) must match the oracle."""
    rng = np.random.default_rng(3)
    entries = []
    n_tok = 40
    for t in range(n_tok):
        for e in range(3):
            docs = np.sort(rng.choice(300, size=120, replace=False)).astype(np.uint64)
            entries.append((t, docs, rng.random(120).astype(np.float32) * 3))
    for thr in (None, 5, 33):
        ids, sc, count = ft.bm25_score(ctx, entries, n_tok, 300.0, 50, thr)
        od, os_, ocount = oracle_topk(entries, n_tok, 300, 50, thr)
        assert count == ocount and ids.tolist() == od.tolist() and np.array_equal(bits(sc), bits(os_))


def test_repeated_docs_inside_one_entry(ctx):
    """BM25Scorer::add is a plain `+=` per call (bm25.rs:369-405): an entry may name the same document several times
    (one field indexed through several locale buckets) and in any order.  Contributions add in entry order and df
    counts the document once — compared bit for bit with the oracle, including the whole score map."""
    rng = np.random.default_rng(11)
    entries = []
    n_tok = 3
    for t in range(n_tok):
        for _ in range(2):
            docs = rng.integers(0, 60, size=200).astype(np.uint64) * np.uint64(7) + np.uint64(3)  # many repeats, unsorted
            entries.append((t, docs, (rng.random(200).astype(np.float32) + np.float32(0.01)) * 2))
    # a run of one doc only, and a doc that appears in every entry
    entries.append((0, np.full(17, 3, dtype=np.uint64), rng.random(17).astype(np.float32)))
    for thr in (None, 2, 3):
        ids, sc, count = ft.bm25_score(ctx, entries, n_tok, 500.0, 40, thr)
        od, os_, ocount = oracle_topk(entries, n_tok, 500, 40, thr)
        assert count == ocount and ids.tolist() == od.tolist() and np.array_equal(bits(sc), bits(os_)), thr
    full = ft.bm25_score_map(ctx, entries, n_tok, 500.0)
    odocs, oscores = orc.search_full_text(entries, n_tok, 500.0, 1.2, None)
    assert sorted(full) == odocs.tolist()
    assert np.array_equal(bits([full[int(d)] for d in odocs]), bits(oscores))


def test_rebuild_forgets_the_old_omc(ctx):
    """A rebuilt store is a new index generation: multipliers set for the previous document table must not leak
    into it (they were indexed by the OLD local document numbers)."""
    docs1 = np.arange(100, dtype=np.uint64)
    lens1 = np.full(100, 5)
    store = ft.PostingsStore(ctx)
    store.build(docs1, [5.0], [ft.PostingList(field=0, docs=docs1, tf=np.arange(1, 101), field_len=lens1)])
    store.set_omc({99: 0.001, 98: 0.002})
    ids, sc, _ = store.search([(0, 0, 1.0)], 1, 100.0, 3)
    assert ids.tolist() == [97, 96, 95]  # 99 and 98 were pushed down
    # rebuild with a different (larger, shifted) document table
    docs2 = np.arange(1000, 1300, dtype=np.uint64)
    tf2 = np.arange(1, 301)
    lens2 = np.full(300, 9)
    store.build(docs2, [9.0], [ft.PostingList(field=0, docs=docs2, tf=tf2, field_len=lens2)])
    ntf = np.array([orc.bm25f_normalized_tf(int(t), 9, 9.0, 0.75) for t in tf2], dtype=np.float32)
    od, os_, ocount = oracle_topk([(0, docs2, ntf)], 1, 300, 20)
    ids, sc, count = store.search([(0, 0, 1.0)], 1, 300.0, 20, apply_omc=True)  # no OMC is set any more
    assert count == ocount and ids.tolist() == od.tolist() and np.array_equal(bits(sc), bits(os_))
    store.set_omc({1299: 0.5})
    ids, sc, _ = store.search([(0, 0, 1.0)], 1, 300.0, 20)
    os2 = orc.apply_omc(*orc.search_full_text([(0, docs2, ntf)], 1, 300.0, 1.2, None), [1299], [0.5])
    td, ts = orc.top_n(docs2, os2, 20)
    assert ids.tolist() == td.tolist() and np.array_equal(bits(sc), bits(ts))
    store.close()


# ----------------------------------------------------------------------------- hybrid
@pytest.mark.parametrize("case", util.load_json("hybrid_kat.json")["cases"], ids=lambda c: c["name"])
def test_hybrid_combine_golden(ctx, case):
    vec = {int(k): v for k, v in case["vec"].items()}
    ftm = {int(k): v for k, v in case["ft"].items()}
    if case["omc"]:
        pytest.skip("OMC on the hybrid path is covered by test_hybrid_resident")
    ids, sc, count = ft.hybrid_combine(ctx, vec, ftm, case["k"])
    assert count == case["count"]
    assert ids.tolist() == case["top_ids"]
    assert bits(sc).tolist() == [e["bits"] for e in case["top_scores"]]


def test_hybrid_resident(ctx, synth):
    """search_hybrid (token_score.rs:357-387) on the resident store: vector map (after the a2 epilogue) +
    BM25F over postings → min-max normalise over both → sum → OMC → count → top-k; bit-exact vs the oracle."""
    meta, fields, doc_ids, allow = synth
    store, list_id = build_store(ctx, meta, fields, doc_ids)
    rng = np.random.default_rng(9)
    for ci in (0, 12, 24, 30):
        case = {**meta["cases"][ci], "filter": False}
        refs = [(ti, list_id[(f, t)], meta["boosts"][f]) for ti, t in enumerate(case["terms"])
                for f in range(meta["n_fields"]) if (f, t) in list_id]
        entries = bm25_synth_entries(meta, fields, case, doc_ids, allow)
        n_tok = len(case["terms"])
        fd, fs = orc.search_full_text(entries, n_tok, float(meta["n_docs"]), 1.2, case["threshold"])
        # vector map: some docs inside the fulltext result, some outside, incl. a negative score
        inside = rng.choice(fd, size=min(5, len(fd)), replace=False) if len(fd) else np.array([], dtype=np.uint64)
        outside = np.setdiff1d(doc_ids[:50], fd)[:4]
        vec = {int(d): float(s) for d, s in zip(np.concatenate([inside, outside]),
                                                 rng.uniform(-0.2, 1.0, size=len(inside) + len(outside)))}
        for omc in (None, {int(doc_ids[3]): 2.0, int(fd[0]) if len(fd) else 1: 0.5}):
            od, os_ = orc.normalize_and_combine(list(vec), list(vec.values()), fd, fs)
            if omc:
                os_ = orc.apply_omc(od, os_, list(omc), list(omc.values()))
                store.set_omc(omc)
            td, ts = orc.top_n(od, os_, 30)
            ids, sc, count = store.search(refs, n_tok, float(meta["n_docs"]), 30, case["threshold"], vector=vec,
                                          apply_omc=omc is not None)
            assert count == len(od)
            assert ids.tolist() == td.tolist(), (ci, omc)
            assert np.array_equal(bits(sc), bits(ts))
    store.close()


def test_rrf_extra(ctx):
    """orama_hybrid_rrf (extra next to the min-max parity path): ranks from the K4 order (score desc, doc asc),
    1/(k + rank) in f32, full-text term first — against a numpy restatement, bit for bit."""
    rng = np.random.default_rng(17)
    ft_docs = rng.choice(5000, size=1200, replace=False)
    ft_map = {int(d): float(np.float32(s)) for d, s in zip(ft_docs, rng.uniform(0, 9, size=1200).round(1))}  # many ties
    vec_docs = np.concatenate([rng.choice(ft_docs, size=30, replace=False), np.arange(6000, 6040)])
    vec_map = {int(d): float(np.float32(s)) for d, s in zip(vec_docs, rng.uniform(0.7, 1.0, size=len(vec_docs)).round(2))}
    for depth, rrf_k, top_k in ((1000, 60.0, 50), (100, 1.0, 4000), (20, 60.0, 10)):
        ids, sc, count = ft.hybrid_rrf(ctx, vec_map, ft_map, top_k, rrf_k=rrf_k, depth=depth)

        def ranked(m):
            return [d for d, _ in sorted(m.items(), key=lambda kv: (-np.float32(kv[1]), kv[0]))][:depth]

        fused = {}
        for r, d in enumerate(ranked(ft_map)):
            fused[d] = F(F(0.0) + F(1.0) / F(F(rrf_k) + F(r + 1)))
        for r, d in enumerate(ranked(vec_map)):
            fused[d] = F(fused.get(d, F(0.0)) + F(1.0) / F(F(rrf_k) + F(r + 1)))
        exp = sorted(fused.items(), key=lambda kv: (-kv[1], kv[0]))[:top_k]
        assert count == len(fused)
        assert ids.tolist() == [d for d, _ in exp]
        assert np.array_equal(bits(sc), bits([s for _, s in exp]))
    ids, sc, count = ft.hybrid_rrf(ctx, {}, {7: 1.0}, 5)
    assert ids.tolist() == [7] and count == 1


def test_batch_entry_equals_single_queries(ctx):
    """orama_post_search_batch: many queries in one call on library worker threads — every result equals the
    single-query call bit for bit (and therefore the oracle, which the other tests of this file pin it to)."""
    rng = np.random.default_rng(19)
    n_docs, n_lists = 30_000, 24
    doc_ids = np.arange(n_docs, dtype=np.uint64) * np.uint64(2) + np.uint64(1)
    lens = rng.integers(5, 150, size=n_docs).astype(np.uint32)
    lists = []
    for l in range(n_lists):
        pos = np.sort(rng.choice(n_docs, size=int(rng.integers(100, 9000)), replace=False))
        lists.append(ft.PostingList(field=0, docs=doc_ids[pos], tf=rng.integers(1, 6, size=len(pos)).astype(np.uint32),
                                    field_len=lens[pos]))
    post = ft.PostingsStore(ctx)
    post.build(doc_ids, [float(lens.mean())], lists)
    queries = []
    for i in range(70):
        nt = int(rng.integers(1, 6))
        ls = rng.choice(n_lists, size=nt, replace=False)
        refs = [(t, int(l), float(rng.choice([1.0, 2.0]))) for t, l in enumerate(ls)]
        queries.append((refs, nt, None if i % 3 else max(1, nt - 1)))
    allow = oa.AllowBitmap(int(doc_ids.max()) + 1, doc_ids[rng.random(n_docs) < 0.8]).to_device(ctx)
    for flt in (None, allow):
        for par in (1, 8):
            got = post.search_batch(queries, float(n_docs), 25, allow=flt, max_parallel=par)
            for (refs, nt, thr), (ids, sc, count) in zip(queries, got):
                e_ids, e_sc, e_count = post.search(refs, nt, float(n_docs), 25, thr, allow=flt)
                assert count == e_count and ids.tolist() == e_ids.tolist()
                assert np.array_equal(sc.view(np.uint32), e_sc.view(np.uint32))
    allow.close()
    post.close()
