"""Facet counting and group-by top-k over the HBM-resident score map (orama_scores / orama_facet_* / orama_group_top,
SURVEY §8f rank 4) against the oracle's restatement of index/facet.rs, index/group.rs and sort.rs:203-213.

Counts are integers: the bar is equality.  Group tops: ids identical, scores bit-identical (they are copies of the map's
scores).  The map itself (every entry, any size) must equal the oracle's search_full_text / normalize_and_combine map
bit for bit — that replaces round 1's 4096-entry cap of get_scores()."""
import numpy as np
import pytest

import oramacore_amd as oa
import util
from oracle import oracle as orc
from oramacore_amd import fulltext as ft
from oramacore_amd.token_score import (FulltextMode, Index, StringFieldStorage, TokenScoreContext, TokenScoreParams,
                                       facets_and_groups)

pytestmark = pytest.mark.gpu
F = np.float32


@pytest.fixture(autouse=True, params=["k3r", "k3"])
def scorer(request, ctx):
    """Every test runs with the score map built by the range scorer (K3r: candidate list + position index out of the
    scoring launch, where the query is eligible) and by the per-record scorer (K3)."""
    ctx.set_bm25_ranges(request.param == "k3r")
    yield request.param
    ctx.set_bm25_ranges(True)


def build_store(ctx, n_docs, n_tok, rng, id_mul=1, id_add=0):
    doc_ids = np.arange(n_docs, dtype=np.uint64) * np.uint64(id_mul) + np.uint64(id_add)
    lens = rng.integers(5, 200, size=n_docs).astype(np.uint32)
    avg = F(lens.mean())
    lists, entries = [], []
    for t in range(n_tok):
        pos = np.sort(rng.choice(n_docs, size=int(rng.integers(n_docs // 10, n_docs // 2)), replace=False))
        tf = rng.integers(1, 5, size=len(pos)).astype(np.uint32)
        lists.append(ft.PostingList(field=0, docs=doc_ids[pos], tf=tf, field_len=lens[pos]))
        ntf = (tf.astype(F) / (F(0.25) + F(0.75) * (lens[pos].astype(F) / avg))).astype(F)
        entries.append((t, doc_ids[pos], ntf))
    post = ft.PostingsStore(ctx)
    post.build(doc_ids, [float(avg)], lists)
    return post, doc_ids, entries


def test_score_map_export_and_lookup_any_size(ctx, scorer):
    rng = np.random.default_rng(1)
    n_docs, n_tok = 60_000, 4  # the map has tens of thousands of entries: far beyond the old 4096 cap
    post, doc_ids, entries = build_store(ctx, n_docs, n_tok, rng, id_mul=3, id_add=11)
    refs = [(t, t, 1.0) for t in range(n_tok)]
    ctx.prof_reset()
    ctx.prof_enable(True)
    sm = post.search_scores(refs, n_tok, float(n_docs), 50)
    ctx.prof_enable(False)
    assert (ctx.prof_get("bm25_range_score")[1] > 0) == (scorer == "k3r") and (ctx.prof_get("bm25_accumulate")[1] > 0) == (scorer == "k3")
    od, os_ = orc.search_full_text(entries, n_tok, float(n_docs), 1.2, None)
    assert len(sm) == len(od) == sm.hits[2] and len(od) > 20_000
    got = sm.to_dict()
    exp = dict(zip(od.tolist(), os_.tolist()))
    assert got.keys() == exp.keys()
    ga = np.array([got[d] for d in od.tolist()], dtype=F)
    assert np.array_equal(ga.view(np.uint32), os_.view(np.uint32))
    # hits of the same call are the top-n of that map
    td, ts = orc.top_n(od, os_, 50)
    assert sm.hits[0].tolist() == td.tolist() and np.array_equal(sm.hits[1].view(np.uint32), ts.view(np.uint32))
    probe = np.concatenate([od[:5], np.array([1, 2, 10**9], dtype=np.uint64)])  # ids not in the index at all
    sc, present = sm.lookup(probe)
    assert present[:5].all() and np.array_equal(sc[:5].view(np.uint32), os_[:5].view(np.uint32))
    assert [bool(p) for p in present[5:]] == [int(d) in exp for d in probe[5:].tolist()]
    sm.close()
    # seam (i): BM25Scorer.get_scores() returns the whole map too
    sc_i = ft.bm25_score_map(ctx, entries, n_tok, float(n_docs))
    assert sc_i.keys() == exp.keys()
    post.close()


@pytest.mark.parametrize("hybrid", [False, True], ids=["fulltext", "hybrid"])
def test_facet_counts_and_group_tops_equal_oracle(ctx, hybrid):
    rng = np.random.default_rng(7)
    n_docs, n_tok = 20_000, 3
    post, doc_ids, entries = build_store(ctx, n_docs, n_tok, rng)
    refs = [(t, t, 1.0) for t in range(n_tok)]
    thr = 2
    vec = None
    od, os_ = orc.search_full_text(entries, n_tok, float(n_docs), 1.2, thr)
    if hybrid:
        vdocs = np.concatenate([rng.choice(od, 5, replace=False), np.setdiff1d(doc_ids[::1234], od)[:5]])
        vec = {int(d): float(s) for d, s in zip(vdocs, rng.uniform(0.1, 1.0, size=len(vdocs)))}
        od, os_ = orc.normalize_and_combine(list(vec), list(vec.values()), od, os_)
    od, os_ = np.asarray(od, dtype=np.uint64), np.asarray(os_, dtype=F)
    sm = post.search_scores(refs, n_tok, float(n_docs), 10, threshold=thr, vector=vec)
    assert len(sm) == len(od)

    # --- bool field (2 buckets, every doc in exactly one) and a string filter field (arrays: a doc in several buckets,
    #     some docs in none, ids the index does not hold)
    is_true = rng.random(n_docs) < 0.3
    bool_buckets = [doc_ids[is_true], doc_ids[~is_true]]
    n_keys = 37
    str_buckets = []
    for kidx in range(n_keys):
        size = int(rng.integers(0, 3000)) if kidx != 5 else 0  # an empty bucket too
        b = np.sort(rng.choice(n_docs, size=size, replace=False)).astype(np.uint64)
        str_buckets.append(np.concatenate([doc_ids[b], np.array([10**8 + kidx], dtype=np.uint64)]))
    for buckets in (bool_buckets, str_buckets):
        fld = ft.FacetField.buckets(post, buckets)
        got = sm.facet_count(fld)
        off = np.concatenate([[0], np.cumsum([len(b) for b in buckets])]).astype(np.uint64)
        exp = orc.facet_count_buckets(od, off, np.concatenate(buckets))
        assert got.tolist() == exp.tolist()
        assert got.sum() > 0
        # group-by over the same buckets: best 7 per group
        g_ids, g_sc, g_n = sm.group_top(fld, 7)
        e_ids, e_sc, e_n = orc.group_top(od, os_, off, np.concatenate(buckets), 7)
        assert g_n.tolist() == e_n.tolist()
        for g in range(len(buckets)):
            m = int(e_n[g])
            assert g_ids[g, :m].tolist() == e_ids[g, :m].tolist(), g
            assert np.array_equal(g_sc[g, :m].view(np.uint32), e_sc[g, :m].view(np.uint32))
        fld.close()

    # --- number field: several numbers per doc, overlapping ranges, inclusive ends, negative / fractional values
    ndocs = np.concatenate([doc_ids, doc_ids[::3], np.array([10**9], dtype=np.uint64)])
    nvals = np.concatenate([rng.integers(-50, 1000, size=n_docs).astype(np.float64),
                            rng.uniform(-5.0, 5.0, size=len(doc_ids[::3])).astype(np.float32).astype(np.float64), [3.0]])
    fld = ft.FacetField.numbers(post, ndocs, nvals)
    ranges = [(0, 100), (100, 100), (50, 500), (-1000, 1000), (2.5, 2.75), (1001, 2000)] + \
             [(i * 10, i * 10 + 9) for i in range(70)]  # > 64 ranges: the library chunks them
    got = sm.facet_count_ranges(fld, ranges)
    exp = orc.facet_count_ranges(od, ndocs, nvals, ranges)
    assert got.tolist() == exp.tolist() and got[3] >= len(od)
    fld.close()
    sm.close()
    post.close()


def test_number_facet_counts_every_stored_number(ctx):
    """The declared rule for number facets (assumption 7): `calculate_facet` is
    `filter(Between).filter(contains_key).count()` with no de-duplication in-tree (number_field.rs:368-387), so a
    document holding two numbers inside one range is counted twice, one number in each of two ranges once in both."""
    d = np.arange(10, dtype=np.uint64)
    post = ft.PostingsStore(ctx)
    post.build(d, [5.0], [ft.PostingList(field=0, docs=d[:6], tf=np.ones(6, dtype=np.uint32), field_len=np.full(6, 5, np.uint32))])
    sm = post.search_scores([(0, 0, 1.0)], 1, 10.0, 10)  # map = docs 0..5
    ndocs = np.array([1, 1, 2, 2, 7, 7, 3], dtype=np.uint64)       # doc 7 is not in the map
    nvals = np.array([10.0, 20.0, 10.0, 500.0, 10.0, 11.0, 100.0])
    fld = ft.FacetField.numbers(post, ndocs, nvals)
    got = sm.facet_count_ranges(fld, [(0, 100), (400, 600), (100, 100), (1000, 2000)])
    assert got.tolist() == [2 + 1 + 1, 1, 1, 0]
    assert got.tolist() == orc.facet_count_ranges(np.arange(6, dtype=np.uint64), ndocs, nvals,
                                                  [(0, 100), (400, 600), (100, 100), (1000, 2000)]).tolist()
    fld.close()
    sm.close()
    post.close()


def test_group_top_large_bucket_and_nan_scores(ctx):
    """One group holding every document (rounds of the LDS-resident running top-k), max_results at the limit, and a
    map whose scores are all NaN (hybrid with max == min): groups come back empty, facet counts still count."""
    rng = np.random.default_rng(3)
    n_docs, n_tok = 30_000, 2
    post, doc_ids, entries = build_store(ctx, n_docs, n_tok, rng)
    refs = [(t, t, 1.0) for t in range(n_tok)]
    sm = post.search_scores(refs, n_tok, float(n_docs), 5)
    od, os_ = orc.search_full_text(entries, n_tok, float(n_docs), 1.2, None)
    fld = ft.FacetField.buckets(post, [doc_ids, doc_ids[:10], np.zeros(0, dtype=np.uint64)])
    off = np.array([0, n_docs, n_docs + 10, n_docs + 10], dtype=np.uint64)
    alld = np.concatenate([doc_ids, doc_ids[:10]])
    for k in (1, 100, 1024):
        g_ids, g_sc, g_n = sm.group_top(fld, k)
        e_ids, e_sc, e_n = orc.group_top(od, os_, off, alld, k)
        assert g_n.tolist() == e_n.tolist() and g_n[0] == min(k, len(od)) and g_n[2] == 0
        for g in range(3):
            m = int(e_n[g])
            assert g_ids[g, :m].tolist() == e_ids[g, :m].tolist(), (k, g)
            assert np.array_equal(g_sc[g, :m].view(np.uint32), e_sc[g, :m].view(np.uint32))
    with pytest.raises(oa.OramaError):
        sm.group_top(fld, 1025)
    fld.close()
    sm.close()
    # NaN scores inside the map (an OMC multiplier of NaN): `count` and the facet counts include those documents
    # (contains_key), top-n and the group tops skip them (NotNan::new -> Err, sort.rs:207, 264-268)
    lens = np.full(100, 10, dtype=np.uint32)
    d100 = np.arange(100, dtype=np.uint64)
    p2 = ft.PostingsStore(ctx)
    p2.build(d100, [10.0], [ft.PostingList(field=0, docs=d100[:40], tf=(1 + d100[:40] % 3).astype(np.uint32),
                                           field_len=lens[:40])])
    p2.set_omc({0: float("nan"), 2: float("nan"), 5: 2.0})
    sm2 = p2.search_scores([(0, 0, 1.0)], 1, 100.0, 50)
    assert sm2.hits[2] == 40 and len(sm2.hits[0]) == 38 and not ({0, 2} & set(sm2.hits[0].tolist()))
    f2 = ft.FacetField.buckets(p2, [d100[:50], d100[30:], d100[:3]])
    assert sm2.facet_count(f2).tolist() == [40, 10, 3]
    g_ids, g_sc, g_n = sm2.group_top(f2, 5)
    assert g_n.tolist() == [5, 5, 1] and g_ids[2, 0] == 1
    m2 = sm2.to_dict()
    assert len(m2) == 40 and np.isnan(m2[0]) and np.isnan(m2[2])
    exp0 = sorted((d for d in range(40) if d not in (0, 2)), key=lambda d: (-float(m2[d]), d))[:5]
    assert g_ids[0].tolist() == exp0
    f2.close()
    sm2.close()
    # a rebuilt index invalidates the field images
    f3 = ft.FacetField.buckets(p2, [d100[:50]])
    p2.build(d100, [10.0], [ft.PostingList(field=0, docs=d100[:40], tf=np.ones(40, dtype=np.uint32), field_len=lens[:40])])
    sm3 = p2.search_scores([(0, 0, 1.0)], 1, 100.0, 10)
    with pytest.raises(oa.OramaError):
        sm3.facet_count(f3)
    sm3.close()
    f3.close()
    p2.close()
    post.close()


def test_mirror_facets_use_the_unfiltered_map_and_groups_the_filtered_one(ctx):
    """search.rs:347-396: with a `where` filter the facets are counted on a re-score WITHOUT it; groups and hits keep
    the filter.  Checked against plain Python set arithmetic over the oracle's maps."""
    n = 300
    colors = ["red", "green", "blue"]
    docs = {i: {"text": "shirt " * (1 + i % 3) + ("cotton " if i % 2 else "wool ") + f"sku{i}"} for i in range(n)}
    idx = Index(ctx)
    idx.string_fields[0] = StringFieldStorage()
    for d, doc in docs.items():
        idx.document_ids.add(d)
        idx.string_fields[0].insert(d, doc["text"])
    idx.bool_fields["in_stock"] = {d: (d % 4 != 0) for d in range(n)}
    idx.string_filter_fields["color"] = {d: ([colors[d % 3], colors[(d + 1) % 3]] if d % 10 == 0 else colors[d % 3])
                                         for d in range(n) if d % 7}
    idx.number_fields["price"] = {d: d % 50 for d in range(n)}
    idx.commit()
    tsc = TokenScoreContext(idx)
    where = oa.AllowBitmap.from_mask(np.arange(n) % 3 == 0)  # the user clicked "red"
    params = TokenScoreParams(mode=FulltextMode("shirt cotton"), limit=10, filtered_doc_ids=where)
    hits, count, facets, groups = facets_and_groups(
        tsc, params, facets={"in_stock": "bool", "color": "string", "price": [(0, 9), (10, 49), (0, 49)], "nope": "bool"},
        group_by=(["color", "in_stock"], 3), has_where_filter=True, not_deleted=None)
    # expected maps from the host postings (prefix semantics of the mirror: "shirt", "cotton" are whole terms here)
    matched_all = {d for d in range(n)}                       # every doc holds "shirt"
    matched_where = {d for d in matched_all if d % 3 == 0}
    assert count == len(matched_where) and all(h[0] % 3 == 0 for h in hits)
    assert "nope" not in facets
    assert facets["in_stock"]["values"] == {"true": sum(1 for d in matched_all if d % 4 != 0),
                                            "false": sum(1 for d in matched_all if d % 4 == 0)}
    exp_color = {c: 0 for c in colors}
    for d, v in idx.string_filter_fields["color"].items():
        for c in (v if isinstance(v, list) else [v]):
            exp_color[c] += 1
    assert facets["color"] == {"count": 3, "values": exp_color}
    assert facets["price"]["values"] == {"0-9": sum(1 for d in range(n) if d % 50 <= 9),
                                         "10-49": sum(1 for d in range(n) if 10 <= d % 50 <= 49), "0-49": n}
    # groups: combinations of (color, in_stock) over the FILTERED map, best 3 by score
    sm = idx._post.search_scores(tsc._refs(["shirt", "cotton"], None, {}, False), 2, float(n), 10, allow=where)
    fmap = sm.to_dict()
    sm.close()
    assert set(fmap) == matched_where
    assert len(groups) == 3 * 2
    for (color, stock), top in groups.items():
        members = [d for d in fmap if d in idx.string_filter_fields["color"]
                   and color in (idx.string_filter_fields["color"][d] if isinstance(idx.string_filter_fields["color"][d], list)
                                 else [idx.string_filter_fields["color"][d]]) and idx.bool_fields["in_stock"][d] is stock]
        exp = sorted(members, key=lambda d: (-float(fmap[d]), d))[:3]
        assert [t[0] for t in top] == exp, (color, stock)
