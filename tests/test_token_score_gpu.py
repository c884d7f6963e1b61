"""End-to-end replicas of the reference's search tests on the scoring dispatcher mirror (GPU).

The reference's tests are ordinal/cardinal (src/tests/fulltext_search.rs, vector_search.rs) and run a real
tokenizer + embedding model; here `SimpleTokenizer` and a deterministic fake `embed` stand in for those
(out-of-scope) components, and every result is additionally compared with the oracle pipeline.
"""
import numpy as np
import pytest

import oramacore_amd as oa
import util
from oracle import oracle as orc
from oramacore_amd.token_score import (DEFAULT_EXACT_MATCH_BOOST, FulltextMode, HybridMode, Index, StringFieldStorage,
                                       TokenScoreContext, TokenScoreParams, VectorMode)

pytestmark = pytest.mark.gpu
F = np.float32


def make_index(ctx, docs: dict, fields=("text",)):
    idx = Index(ctx)
    for fi, name in enumerate(fields):
        idx.string_fields[fi] = StringFieldStorage()
    for doc_id, doc in docs.items():
        idx.document_ids.add(doc_id)
        for fi, name in enumerate(fields):
            if name in doc:
                idx.string_fields[fi].insert(doc_id, doc[name])
    idx.commit()
    return idx


def oracle_fulltext(idx: Index, tokens, exact, boost=None, threshold=None, allow=None):
    """The same query through the oracle: entries built on the host from the index's postings."""
    entries = []
    for ti, tok in enumerate(tokens):
        for fid in sorted(idx.string_fields):
            sf = idx.string_fields[fid]
            terms = [tok] if exact else [t for t in sorted(sf.postings) if t.startswith(tok)]
            for term in terms:
                if term not in sf.postings:
                    continue
                pl = sorted(sf.postings[term].items())
                if allow is not None:
                    pl = [(d, tf) for d, tf in pl if allow.contains(d)]
                docs = [d for d, _ in pl]
                bo = F((boost or {}).get(fid, 1.0))
                if term == tok:  # the exact-match factor of the store (the mirror's declared default, > 1)
                    bo = F(bo * F(DEFAULT_EXACT_MATCH_BOOST))
                ntf = [F(bo * orc.bm25f_normalized_tf(tf, sf.field_len[d], sf.avg_field_length(), 0.75)) for d, tf in pl]
                entries.append((ti, docs, ntf))
    return orc.search_full_text(entries, len(tokens), float(idx.document_count), 1.2, threshold)


def test_search_documents_order(ctx):
    """src/tests/fulltext_search.rs:146-189 — the shorter document ranks first."""
    idx = make_index(ctx, {1: {"text": "This is a long text with a lot of words"}, 2: {"text": "This is a smaller text"}})
    hits, count = TokenScoreContext(idx).execute(TokenScoreParams(mode=FulltextMode("text")))
    assert count == 2 and [h[0] for h in hits] == [2, 1] and hits[0][1] > hits[1][1]


def test_fulltext_threshold(ctx):
    """src/tests/fulltext_search.rs:478-600 — threshold = fraction of query tokens a document must match."""
    idx = make_index(ctx, {1: {"text": "The pen is on the table"},
                           2: {"text": "the pen", "text2": "is on the table"},
                           3: {"text": "the pen"}}, fields=("text", "text2"))
    tsc = TokenScoreContext(idx)

    def n_hits(term, thr):
        hits, count = tsc.execute(TokenScoreParams(mode=FulltextMode(term, threshold=thr)))
        assert count == len(hits)
        return len(hits)

    assert n_hits("the pen is on the table", 0.7) == 2
    assert n_hits("the pen is on the table", 1.0) == 2
    assert n_hits("pen", 0.0) == 3
    assert n_hits("pen", 1.0) == 3


def test_prefix_vs_exact(ctx):
    """src/tests/fulltext_search.rs:603-753 — "christoph" matches "Christopher" unless `exact`."""
    idx = make_index(ctx, {1: {"text": "Christopher Nolan"}, 2: {"text": "Christoph Waltz"}, 3: {"text": "Someone else"}})
    tsc = TokenScoreContext(idx)
    hits, count = tsc.execute(TokenScoreParams(mode=FulltextMode("christoph")))
    assert count == 2 and {h[0] for h in hits} == {1, 2}
    hits, count = tsc.execute(TokenScoreParams(mode=FulltextMode("christoph", exact=True)))
    assert count == 1 and hits[0][0] == 2
    # the exact token outranks the prefix match (src/tests/boost_integration.rs:449-491 in spirit)
    hits, _ = tsc.execute(TokenScoreParams(mode=FulltextMode("christoph")))
    od, os_ = oracle_fulltext(idx, ["christoph"], exact=False)
    td, ts = orc.top_n(od, os_, 10)
    assert [h[0] for h in hits] == td.tolist()
    assert np.array_equal(np.array([h[1] for h in hits], dtype=np.float32).view(np.uint32), ts.view(np.uint32))


def test_empty_term_and_paging(ctx):
    idx = make_index(ctx, {i: {"text": "text " * (i + 1)} for i in range(100)})
    tsc = TokenScoreContext(idx)
    # src/tests/fulltext_search.rs:192-251: 99, 98, … with limit 10, count 100
    hits, count = tsc.execute(TokenScoreParams(mode=FulltextMode("text"), limit=10))
    assert count == 100 and [h[0] for h in hits] == list(range(99, 89, -1))
    # :254-335 offset paging
    hits, count = tsc.execute(TokenScoreParams(mode=FulltextMode("text"), limit=5, offset=20))
    assert count == 100 and [h[0] for h in hits] == [79, 78, 77, 76, 75]
    # unknown term: nothing
    hits, count = tsc.execute(TokenScoreParams(mode=FulltextMode("absent")))
    assert count == 0 and hits == []


@pytest.mark.parametrize("n", [1000, 5000])
def test_fulltext_simple_bench_workload(ctx, n):
    """BASELINE configs[0] (plumbing): the benches/fulltext_simple.rs:383-400 corpus and queries
    ("technology", "technology software", "development", limit 10) — results identical to the oracle."""
    docs = {i: {"text": f"document content technology software development number {i}"} for i in range(n)}
    idx = make_index(ctx, docs)
    tsc = TokenScoreContext(idx)
    for term in ("technology", "technology software", "development", "number 7"):
        hits, count = tsc.execute(TokenScoreParams(mode=FulltextMode(term), limit=10))
        toks = [t for t, _ in tsc.text_parser.tokenize_and_stem(term)]
        od, os_ = oracle_fulltext(idx, toks, exact=False)
        td, ts = orc.top_n(od, os_, 10)
        assert count == len(od)
        assert [h[0] for h in hits] == td.tolist(), term
        assert np.array_equal(np.array([h[1] for h in hits], dtype=np.float32).view(np.uint32), ts.view(np.uint32))
    assert count >= 1


def test_boost_and_filter_and_omc(ctx):
    docs = {i: {"title": f"alpha beta {'gamma ' * (i % 3)}", "body": f"beta {'alpha ' * (i % 5)} delta"} for i in range(200)}
    idx = make_index(ctx, docs, fields=("title", "body"))
    idx.omc = {3: 2.0, 7: 0.25, 150: 10.0}
    idx.commit()
    tsc = TokenScoreContext(idx)
    allow = oa.AllowBitmap.from_mask(np.arange(200) % 4 != 1)
    for boost in ({}, {0: 2.0}, {1: 0.5, 0: 3.0}):
        for flt in (None, allow):
            hits, count = tsc.execute(TokenScoreParams(mode=FulltextMode("alpha gamma"), boost=boost, limit=15,
                                                       filtered_doc_ids=flt))
            od, os_ = oracle_fulltext(idx, ["alpha", "gamma"], exact=False, boost=boost, allow=flt)
            os_ = orc.apply_omc(od, os_, list(idx.omc), list(idx.omc.values()))
            td, ts = orc.top_n(od, os_, 15)
            assert count == len(od) and [h[0] for h in hits] == td.tolist()
            assert np.array_equal(np.array([h[1] for h in hits], dtype=np.float32).view(np.uint32), ts.view(np.uint32))


def fake_embed(dim):
    def embed(term: str, model):
        seed = sum(ord(c) * (i + 1) for i, c in enumerate(term)) % 100000
        return util.gaussian_rows(1, dim, seed=seed, scale_rows=False)[0]
    return embed


def test_vector_and_hybrid_modes(ctx):
    """search_vector + search_hybrid (token_score.rs:309-387): a fake deterministic embedder, multi-row docs
    (chunked long text → several vectors per doc, src/tests/vector_search.rs:638-679), similarity cut-off, then
    the full oracle pipeline: scan → epilogue → min-max combine with BM25F → OMC → top-n."""
    dim, n = 384, 400
    docs = {i: {"text": ("red " * (i % 4 + 1)) + ("blue " if i % 3 else "") + f"item{i}"} for i in range(n)}
    idx = make_index(ctx, docs)
    ef = oa.EmbeddingFieldStorage(ctx, oa.Model.BGESmall)
    embed = fake_embed(dim)
    q = embed("red blue", None)
    rng = np.random.default_rng(5)
    rows, row_doc = [], []
    for i in range(n):
        for c in range(1 + i % 3):  # 1-3 chunks per doc
            noise = rng.standard_normal(dim).astype(np.float32)
            w = np.float32(0.95 if i % 10 == 0 else 0.2)
            rows.append(w * q / np.linalg.norm(q) + (1 - w) * noise / np.linalg.norm(noise))
            row_doc.append(i)
    rows = np.ascontiguousarray(rows, dtype=np.float32)
    row_doc = np.array(row_doc, dtype=np.uint64)
    ef.insert_rows(row_doc, rows)
    idx.embedding_fields[0] = ef
    idx.omc = {0: 3.0, 10: 0.5}
    idx.commit()
    tsc = TokenScoreContext(idx, embed=embed)

    # vector mode: hits are docs, chunk scores summed; cut-off 0.7 keeps only the near-duplicates
    # (limit counts ROWS, like the reference's storage: 200 covers every chunk of the ~40 near-duplicate docs)
    hits, count = tsc.execute(TokenScoreParams(mode=VectorMode("red blue", similarity=0.7), limit=200))
    o_ids, o_dist, _ = orc.vector_search(rows, row_doc, q, 200)
    omap = orc.embedding_epilogue(o_ids, o_dist, False, 0.7)
    assert count == len(omap) and {h[0] for h in hits} == set(omap)
    assert all(h[0] % 10 == 0 for h in hits)
    for d, s in hits:
        exp = float(omap[d]) * idx.omc.get(d, 1.0)
        assert abs(s - exp) <= 1e-3

    # hybrid mode, bit-exact combine given the device's own vector map
    for sim in (0.0, 0.7):
        params = TokenScoreParams(mode=HybridMode("red blue", similarity=sim), limit=20)
        hits, count = tsc.execute(params)
        vec = tsc.search_vector(params.mode, params)
        fd, fs = oracle_fulltext(idx, ["red", "blue"], exact=False)
        cd, cs = orc.normalize_and_combine(list(vec), list(vec.values()), fd, fs)
        cs = orc.apply_omc(cd, cs, list(idx.omc), list(idx.omc.values()))
        td, ts = orc.top_n(cd, cs, 20)
        assert count == len(cd) and [h[0] for h in hits] == td.tolist()
        assert np.array_equal(np.array([h[1] for h in hits], dtype=np.float32).view(np.uint32), ts.view(np.uint32))
    ef.close()


def test_fused_hybrid_search_equals_two_call_path(ctx):
    """orama_hybrid_search (two HIP streams, epilogue inside the library) == vector search + host epilogue +
    orama_post_search_hybrid, for BGE and E5 (rescale) models, with and without a filter / OMC."""
    dim, n = 384, 600
    docs = {i: {"text": ("red " * (i % 4 + 1)) + ("blue " if i % 3 else "") + f"item{i}"} for i in range(n)}
    idx = make_index(ctx, docs)
    rng = np.random.default_rng(11)
    q = util.gaussian_rows(1, dim, seed=77, scale_rows=False)[0]
    rows, row_doc = [], []
    for i in range(n):
        for c in range(1 + i % 3):
            noise = rng.standard_normal(dim).astype(np.float32)
            w = np.float32(0.97 if i % 7 == 0 else 0.3)
            rows.append(w * q / np.linalg.norm(q) + (1 - w) * noise / np.linalg.norm(noise))
            row_doc.append(i)
    rows = np.ascontiguousarray(rows, dtype=np.float32)
    row_doc = np.array(row_doc, dtype=np.uint64)
    idx.omc = {0: 3.0, 7: 0.5, 14: 2.0}
    idx.commit()
    allow = oa.AllowBitmap.from_mask(np.arange(n) % 5 != 2)
    tsc = TokenScoreContext(idx)
    refs = tsc._refs(["red", "blue"], None, {}, False)
    for model in (oa.Model.BGESmall, oa.Model.MultilingualE5Small):
        ef = oa.EmbeddingFieldStorage(ctx, model)
        ef.insert_rows(row_doc, rows)
        for sim in (0.0, 0.7):
            for flt in (None, allow):
                for limit in (10, 300):
                    vec = {}
                    ef.search(oa.VectorSearchParams(target=q, similarity=sim, limit=limit, filtered_doc_ids=flt), vec)
                    e_ids, e_sc, e_cnt = idx._post.search(refs, 2, float(n), 25, None, allow=flt, vector=vec)
                    f_ids, f_sc, f_cnt = idx._post.hybrid_search(ef, q, limit, sim, refs, 2, float(n), 25, None,
                                                                 allow=flt, rescale_e5=model.is_e5())
                    assert f_cnt == e_cnt and f_ids.tolist() == e_ids.tolist(), (model, sim, limit)
                    assert np.array_equal(f_sc.view(np.uint32), e_sc.view(np.uint32))
        ef.close()


def test_fulltext_tolerance(ctx):
    """src/tests/fulltext_search.rs:956-1018 — "Mxin" / "Msple" with tolerance 1 find exactly one document each."""
    idx = make_index(ctx, {1: {"text": "Main Street"}, 2: {"text": "Maple Avenue"}, 3: {"text": "Another Street"}})
    tsc = TokenScoreContext(idx)
    hits, count = tsc.execute(TokenScoreParams(mode=FulltextMode("Mxin", tolerance=1)))
    assert count == 1 and [h[0] for h in hits] == [1]
    hits, count = tsc.execute(TokenScoreParams(mode=FulltextMode("Msple", tolerance=1)))
    assert count == 1 and [h[0] for h in hits] == [2]
    hits, count = tsc.execute(TokenScoreParams(mode=FulltextMode("Mxin")))  # no tolerance: nothing
    assert count == 0 and hits == []
    hits, count = tsc.execute(TokenScoreParams(mode=FulltextMode("street", tolerance=1)))
    assert count == 2 and sorted(h[0] for h in hits) == [1, 3]


def test_boost_ratio_bounds(ctx):
    """src/tests/boost_integration.rs:370-447 in spirit: raising a field's boost raises the score of documents
    matching in that field monotonically, sub-linearly (BM25 saturation k = 1.2), and leaves the ordering among
    them intact; results equal the oracle bit for bit."""
    docs = {i: {"title": "gpu search " + "pad " * (i % 7), "body": "search engine " + "gpu " * (i % 3)} for i in range(60)}
    idx = make_index(ctx, docs, fields=("title", "body"))
    tsc = TokenScoreContext(idx)
    prev = None
    for boost in (1.0, 2.0, 4.0):
        hits, count = tsc.execute(TokenScoreParams(mode=FulltextMode("gpu"), boost={0: boost}, limit=60))
        od, os_ = oracle_fulltext(idx, ["gpu"], exact=False, boost={0: boost})
        td, ts = orc.top_n(od, os_, 60)
        assert count == len(od) and [h[0] for h in hits] == td.tolist()
        assert np.array_equal(np.array([h[1] for h in hits], dtype=np.float32).view(np.uint32), ts.view(np.uint32))
        sc = dict(hits)
        if prev is not None:
            for d in sc:
                assert sc[d] > prev[d]                       # monotone
                assert sc[d] / prev[d] < 2.0                 # saturating: doubling the boost less than doubles the score
            assert min(sc[d] / prev[d] for d in sc) > 1.05
        prev = sc


def test_search_on_several_indexes(ctx):
    """search.rs:297-343, 481-500: two indexes of one collection (disjoint DocumentIds, their own BM25 statistics
    and OMC maps) — the merged result equals the oracle's per-index maps extended into one, then top-n."""
    from oramacore_amd.token_score import search_on_indexes

    docs_a = {i: {"text": "alpha beta " + "gamma " * (i % 4)} for i in range(0, 120)}
    docs_b = {i: {"text": "beta gamma delta " + "alpha " * (i % 3)} for i in range(1000, 1090)}
    idx_a, idx_b = make_index(ctx, docs_a), make_index(ctx, docs_b)
    idx_b.omc = {1003: 5.0, 1010: 0.5}
    idx_b.commit()
    tscs = [TokenScoreContext(idx_a), TokenScoreContext(idx_b)]
    for query, limit, offset in (("alpha gamma", 10, 0), ("beta", 25, 5), ("delta", 10, 0), ("absent", 10, 0)):
        hits, count = search_on_indexes(tscs, TokenScoreParams(mode=FulltextMode(query), limit=limit, offset=offset))
        toks = [t for t, _ in tscs[0].text_parser.tokenize_and_stem(query)]
        all_d, all_s = [], []
        for idx in (idx_a, idx_b):
            od, os_ = oracle_fulltext(idx, toks, exact=False)
            if idx.omc:
                os_ = orc.apply_omc(od, os_, list(idx.omc), list(idx.omc.values()))
            all_d.append(od)
            all_s.append(os_)
        od, os_ = np.concatenate(all_d), np.concatenate(all_s)
        td, ts = orc.top_n(od, os_, limit + offset)
        assert count == len(od)
        assert [h[0] for h in hits] == td[offset:].tolist(), query
        assert np.array_equal(np.array([h[1] for h in hits], dtype=np.float32).view(np.uint32), ts[offset:].view(np.uint32))


def test_hybrid_on_several_indexes_with_offset(ctx):
    """search.rs:334 `limit_hint: score_params.limit` → token_score.rs:339-344/488/497: with offset > 0 the vector
    leg of every index still asks the storage for `limit` rows (NOT limit + offset); only the final cut uses
    limit + offset.  Two indexes, hybrid mode, offset 7: candidate set, count, per-index min/max and therefore every
    score must equal the oracle pipeline run with k = limit."""
    from oramacore_amd.token_score import search_on_indexes

    dim = 384
    embed = fake_embed(dim)
    q = embed("red blue", None)
    rng = np.random.default_rng(21)
    tscs, per_index = [], []
    for base, n in ((0, 150), (5000, 90)):
        docs = {base + i: {"text": ("red " * (i % 3 + 1)) + ("blue " if i % 2 else "") + f"w{i}"} for i in range(n)}
        idx = make_index(ctx, docs)
        ef = oa.EmbeddingFieldStorage(ctx, oa.Model.BGESmall)
        rows = []
        for i in range(n):
            noise = rng.standard_normal(dim).astype(np.float32)
            w = np.float32(0.9 - 0.004 * i)
            rows.append(w * q / np.linalg.norm(q) + (1 - w) * noise / np.linalg.norm(noise))
        rows = np.ascontiguousarray(rows, dtype=np.float32)
        row_doc = np.arange(base, base + n, dtype=np.uint64)
        ef.insert_rows(row_doc, rows)
        idx.embedding_fields[0] = ef
        tscs.append(TokenScoreContext(idx, embed=embed))
        per_index.append((idx, ef, rows, row_doc))
    limit, offset = 6, 7
    hits, count = search_on_indexes(tscs, TokenScoreParams(mode=HybridMode("red blue", similarity=0.0), limit=limit,
                                                           offset=offset))
    all_d, all_s = [], []
    for idx, ef, rows, row_doc in per_index:
        o_ids, o_dist, _ = orc.vector_search(rows, row_doc, q, limit)      # k = limit, not limit + offset
        vmap = orc.embedding_epilogue(o_ids, o_dist, False, 0.0)
        fd, fs = oracle_fulltext(idx, ["red", "blue"], exact=False)
        cd, cs = orc.normalize_and_combine(list(vmap), list(vmap.values()), fd, fs)
        all_d.append(np.asarray(cd, dtype=np.uint64))
        all_s.append(np.asarray(cs, dtype=np.float32))
    od, os_ = np.concatenate(all_d), np.concatenate(all_s)
    td, ts = orc.top_n(od, os_, limit + offset)
    assert count == len(od)
    assert [h[0] for h in hits] == td[offset:].tolist()
    got = np.array([h[1] for h in hits], dtype=np.float32)
    # the vector similarities come from the device scan (tree-reduced dot products): 1e-4 on the cosine, the
    # min-max normalisation keeps that scale
    assert np.max(np.abs(got - ts[offset:])) <= 2e-4
    for _, ef, _, _ in per_index:
        ef.close()
