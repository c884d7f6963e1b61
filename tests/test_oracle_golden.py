"""CPU tests: the C oracle (oracle/orama_oracle.c) against the golden fixtures.

bm25_kat.json holds the reference's own known-answer tests (src/collection_manager/bm25.rs:533-1043);
the other fixtures come from the independent numpy restatement in tests/golden/make_golden.py.
"""
import numpy as np
import pytest

import util
from oracle import oracle as orc

F = np.float32


# ----------------------------------------------------------------------------- BM25F known answers
def _run_kat(case):
    """Evaluate one bm25.rs test case with the oracle; returns {doc: score}."""
    scores = {}
    if case["kind"] == "legacy_add":
        for a in case["adds"]:
            v = orc.bm25_legacy_add(a["tf"], a["len"], a["avglen"], a["total_docs"], a["df"], a["k"],
                                    a["weight"], a["b"], a["boost"])
            if not np.isnan(v):
                scores[a["doc"]] = F(scores.get(a["doc"], F(0.0)) + v)
    else:  # add_field + finalize_term (bm25.rs:942-952)
        fin = case["finalize"]
        docs, ntfs = [], []
        for f in case["fields"]:
            ntf = orc.bm25f_normalized_tf(f["tf"], f["len"], f["avglen"], f["b"])
            docs.append(f["doc"])
            ntfs.append(F(F(f["weight"]) * ntf))  # contrib.weight * contrib.normalized_tf
        # all fields of the case belong to one token; df / N / k come from finalize_term's arguments:
        # feed them through the scalar helpers to keep df independent of the posting count.
        idf = orc.bm25_idf(fin["total_docs"], fin["df"])
        by_doc = {}
        for d, v in zip(docs, ntfs):
            by_doc.setdefault(d, []).append(v)
        for d, vs in by_doc.items():
            s = F(0.0)
            for v in vs:
                s = F(s + v)
            scores[d] = orc.bm25f_score(s, fin["k"], idf)
    return scores


@pytest.mark.parametrize("case", util.load_json("bm25_kat.json")["cases"], ids=lambda c: c["name"])
def test_bm25_reference_known_answers(case):
    scores = _run_kat(case)
    for doc, exp in case.get("expect", {}).items():
        assert abs(float(scores[int(doc)]) - exp) <= case["tol"], (case["name"], scores, exp)
    for rel in case.get("relations", []):
        op = rel[0]
        if op == "gt":
            assert scores[rel[1]] > scores[rel[2]]
        elif op == "lt":
            assert scores[rel[1]] < scores[rel[2]]
        elif op == "gt_value":
            assert scores[rel[1]] > rel[2]
        elif op == "le_value":
            assert scores[rel[1]] <= rel[2]
        elif op == "ratio_gt":
            assert scores[rel[1]] / scores[rel[2]] > rel[3]
        elif op == "ratio_lt":
            assert scores[rel[1]] / scores[rel[2]] < rel[3]
        else:
            raise AssertionError(op)
    if "n_docs" in case:
        assert len(scores) == case["n_docs"]


def test_bm25_add_field_path_matches_full_text_loop():
    """The add_field + finalize KAT through orc_search_full_text (weight folded into ntf)."""
    case = [c for c in util.load_json("bm25_kat.json")["cases"]
            if c["name"] == "test_canonical_bm25f_single_term_two_fields"][0]
    # search_full_text derives df from the postings: build 10 docs containing the term so df == 10
    docs0, ntf0, docs1, ntf1 = [], [], [], []
    f0, f1 = case["fields"]
    for d in range(1, 11):
        docs0.append(d)
        ntf0.append(F(F(f0["weight"]) * orc.bm25f_normalized_tf(f0["tf"], f0["len"], f0["avglen"], f0["b"])))
    docs1.append(1)
    ntf1.append(F(F(f1["weight"]) * orc.bm25f_normalized_tf(f1["tf"], f1["len"], f1["avglen"], f1["b"])))
    docs, scores = orc.search_full_text([(0, docs0, ntf0), (0, docs1, ntf1)], 1, 100.0, 1.2)
    got = dict(zip(docs.tolist(), scores.tolist()))
    assert abs(got[1] - case["expect"]["1"]) <= case["tol"]
    assert len(got) == 10


# ----------------------------------------------------------------------------- cosine
@pytest.mark.parametrize("d", [384, 768])
def test_cosine_small_golden(d):
    g = np.load(util.GOLDEN / "cosine_small.npz")
    n, nq, k = 4096, 8, 100
    corpus = util.det_matrix(n, d, seed=1000 + d)
    queries = util.det_matrix(nq, d, seed=2000 + d)
    row_doc = np.arange(n, dtype=np.uint64)
    for qi in range(nq):
        ids, dist, rows = orc.vector_search(corpus, row_doc, queries[qi], k)
        assert np.array_equal(ids, g[f"ids_{d}"][qi])
        assert np.array_equal(dist, g[f"dist32_{d}"][qi])  # same sequential f32 arithmetic: bit-equal
        assert np.max(np.abs(dist.astype(np.float64) - g[f"dist64_{d}"][qi])) < 1e-5


def test_cosine_ties_golden():
    g = util.load_json("cosine_ties.json")
    d, base_n = 384, 64
    base = util.det_matrix(base_n, d, seed=77)
    order = np.argsort(util.hash_u64(np.arange(base_n * 4, dtype=np.uint64) + np.uint64(5)), kind="stable")
    corpus = np.concatenate([base] * 4, axis=0)[order]
    row_doc = (np.arange(base_n * 4, dtype=np.uint64) * np.uint64(3) + np.uint64(7))[
        np.argsort(util.hash_u64(np.arange(base_n * 4, dtype=np.uint64) + np.uint64(11)), kind="stable")]
    q = util.det_matrix(1, d, seed=78)[0]
    for k, exp in g.items():
        ids, dist, rows = orc.vector_search(corpus, row_doc, q, int(k))
        assert ids.tolist() == exp["ids"]
        assert rows.tolist() == exp["rows"]
        assert dist.view(np.uint32).tolist() == exp["dist_bits"]


def multirow_inputs():
    d, n_docs = 384, 300
    rows_per = (util.hash_u64(np.arange(n_docs, dtype=np.uint64) + np.uint64(123)) % np.uint64(5)).astype(int) + 1
    row_doc = np.repeat(np.arange(n_docs, dtype=np.uint64) + np.uint64(1000), rows_per)
    n = int(row_doc.shape[0])
    corpus = util.det_matrix(n, d, seed=99)
    q = util.det_matrix(1, d, seed=98)[0]
    for j, r in enumerate(range(0, n, 37)):
        alpha = F(0.5 + 0.05 * j)
        corpus[r] = (alpha * q + (F(1.0) - alpha) * corpus[r]).astype(np.float32)
    dead_docs = {1003, 1050, 1100}
    dead = np.array([int(x) in dead_docs for x in row_doc], dtype=np.uint8)
    allow = np.zeros(1400, dtype=bool)
    allow[1000:1300:2] = True
    allow[1001] = True
    return corpus, row_doc, q, dead, allow, dead_docs


def test_cosine_multirow_golden():
    g = util.load_json("cosine_multirow.json")
    corpus, row_doc, q, dead, allow, _ = multirow_inputs()
    assert g["n_rows"] == corpus.shape[0]
    words = np.zeros((allow.size + 63) // 64, dtype=np.uint64)
    for i in np.nonzero(allow)[0]:
        words[i >> 6] |= np.uint64(1) << np.uint64(i & 63)
    for name, kw in (("plain", {}), ("dead", {"dead": dead}),
                     ("filter", {"allow_words": words, "allow_bits": allow.size}),
                     ("dead_filter", {"dead": dead, "allow_words": words, "allow_bits": allow.size})):
        for k in (5, 10, 50):
            exp = g[f"{name}_k{k}"]
            ids, dist, rows = orc.vector_search(corpus, row_doc, q, k, **kw)
            assert ids.tolist() == exp["ids"]
            assert rows.tolist() == exp["rows"]
            assert np.allclose(dist, np.array(exp["dist"], dtype=np.float32), atol=0, rtol=0)
            for is_e5 in (0, 1):
                for ms in (0.0, 0.7):
                    m = orc.embedding_epilogue(ids, dist, bool(is_e5), ms)
                    expm = exp[f"map_e5{is_e5}_min{ms}"]
                    assert {str(dk): float(v) for dk, v in sorted(m.items())} == expm


def test_row_validity_rule():
    assert orc.row_is_valid(np.ones(8, dtype=np.float32))
    assert not orc.row_is_valid(np.zeros(8, dtype=np.float32))
    bad = np.ones(8, dtype=np.float32)
    bad[3] = np.nan
    assert not orc.row_is_valid(bad)
    bad[3] = np.inf
    assert not orc.row_is_valid(bad)


def test_rescale_score_e5():
    assert orc.rescale_score(0.5, False) == F(0.5)
    assert orc.rescale_score(0.5, True) == F(0.0)
    assert orc.rescale_score(1.5, True) == F((F(1.0) - F(0.7)) / F(F(1.0) - F(0.7)))
    v = orc.rescale_score(0.85, True)
    assert v == F(F(F(0.85) - F(0.7)) / F(F(1.0) - F(0.7)))


# ----------------------------------------------------------------------------- synthetic BM25
def bm25_synth_entries(meta, fields, case, doc_ids, allow):
    entries = []
    for ti, term in enumerate(case["terms"]):
        for f in range(meta["n_fields"]):
            docs, ntfs = [], []
            for dix, tf in fields[f]["postings"].get(term, []):
                if case["filter"] and not allow[dix]:
                    continue
                v = F(F(meta["boosts"][f]) * orc.bm25f_normalized_tf(tf, int(fields[f]["lens"][dix]),
                                                                      fields[f]["avg"], meta["b"]))
                docs.append(int(doc_ids[dix]))
                ntfs.append(v)
            entries.append((ti, docs, ntfs))
    return entries


@pytest.fixture(scope="module")
def bm25_synth():
    meta = util.load_json("bm25_synth.json")
    fields = util.mg.zipf_corpus(meta["n_docs"], meta["vocab"], meta["n_fields"], seed=meta["seed"])
    doc_ids = np.arange(meta["n_docs"], dtype=np.uint64) * np.uint64(meta["doc_id_mul"]) + np.uint64(meta["doc_id_add"])
    allow = (util.hash_u64(doc_ids + np.uint64(5)) % np.uint64(3)) != 0
    return meta, fields, doc_ids, allow


def test_bm25_synth_golden(bm25_synth):
    meta, fields, doc_ids, allow = bm25_synth
    for case in meta["cases"]:
        entries = bm25_synth_entries(meta, fields, case, doc_ids, allow)
        docs, scores = orc.search_full_text(entries, len(case["terms"]), float(meta["n_docs"]), 1.2,
                                            case["threshold"])
        assert len(docs) == case["count"]
        tdocs, tscores = orc.top_n(docs, scores, 20)
        assert np.allclose(tscores, np.array(case["top_scores"], dtype=np.float32), rtol=2e-6, atol=1e-6)
        # ids must agree wherever the golden scores are separated by more than the comparison noise
        exp = case["top_scores"]
        for i, (a, b) in enumerate(zip(tdocs.tolist(), case["top_ids"])):
            isolated = (i == 0 or exp[i - 1] - exp[i] > 1e-5) and (i == len(exp) - 1 or exp[i] - exp[i + 1] > 1e-5)
            if isolated:
                assert a == b, (case, i)
        assert abs(float(np.sum(scores.astype(np.float64))) - case["checksum"]) <= 1e-4 * max(1.0, case["checksum"])


def test_fulltext_ordinal_replica():
    """src/tests/fulltext_search.rs:192-251 — ranking 99, 98, 97 …, count 100."""
    g = util.load_json("fulltext_ordinal.json")
    n = g["n"]
    ntfs = [orc.bm25f_normalized_tf(L, L, g["avg"], 0.75) for L in range(1, n + 1)]
    docs, scores = orc.search_full_text([(0, list(range(n)), ntfs)], 1, float(n))
    assert len(docs) == g["count"] == 100
    tdocs, tscores = orc.top_n(docs, scores, 10)
    assert tdocs.tolist() == g["top_ids"] == list(range(99, 89, -1))
    assert np.allclose(tscores, g["top_scores"], rtol=1e-6)


# ----------------------------------------------------------------------------- hybrid / OMC / top-n
def _bits(x):
    return int(np.asarray(np.float32(x)).view(np.uint32))


@pytest.mark.parametrize("case", util.load_json("hybrid_kat.json")["cases"], ids=lambda c: c["name"])
def test_hybrid_and_omc_golden(case):
    vec = {int(k): v for k, v in case["vec"].items()}
    ft = {int(k): v for k, v in case["ft"].items()}
    docs, scores = orc.normalize_and_combine(list(vec), list(vec.values()), list(ft), list(ft.values()))
    if case["omc"]:
        omc = {int(k): v for k, v in case["omc"].items()}
        scores = orc.apply_omc(docs, scores, list(omc), list(omc.values()))
    assert len(docs) == case["count"]
    for d, s in zip(docs.tolist(), scores.tolist()):
        exp = case["combined"][str(d)]
        if exp["v"] is None:
            assert np.isnan(s)
        else:
            assert _bits(s) == exp["bits"], (case["name"], d, s, exp)
    tdocs, tscores = orc.top_n(docs, scores, case["k"])
    assert tdocs.tolist() == case["top_ids"]
    assert [_bits(s) for s in tscores] == [e["bits"] for e in case["top_scores"]]


def test_top_n_tie_rule_and_nan():
    doc = np.array([9, 3, 7, 1, 5, 2], dtype=np.uint64)
    score = np.array([1.0, 2.0, np.nan, 2.0, 1.0, -0.0], dtype=np.float32)
    d, s = orc.top_n(doc, score, 4)
    assert d.tolist() == [1, 3, 5, 9]
    d, s = orc.top_n(doc, score, 10)
    assert d.tolist() == [1, 3, 5, 9, 2]  # NaN dropped
