#!/usr/bin/env python3
"""Generator of the golden fixtures under tests/golden/.

What these fixtures are (and are not):
  * bm25_kat.json transcribes the INPUTS and the closed-form EXPECTED expressions of the reference's
    own known-answer tests (src/collection_manager/bm25.rs:533-1043) as data; the expected numbers
    are evaluated here in numpy float32 exactly as the Rust test bodies write them.  This is the
    only part of the path the reference pins numerically.
  * everything else (cosine, ties, multi-row, hybrid, OMC, synthetic BM25, ordinal replicas) is
    produced by the numpy restatement in THIS file — a second, independent restatement of the
    semantics in SURVEY.md §8, used to cross-check the C oracle (oracle/orama_oracle.c) and the HIP
    path.  The reference cannot be run in the build container (no Rust toolchain, crates not
    vendored), so these are "parity unpinned" goldens: they pin OUR declared semantics.

Run:  python tests/golden/make_golden.py      (rewrites the fixtures; deterministic)
"""
from __future__ import annotations

import json
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
F = np.float32


# ----------------------------------------------------------------------------- deterministic data
def hash_u64(a: np.ndarray) -> np.ndarray:
    """splitmix64 finaliser on uint64 arrays (wrapping arithmetic)."""
    a = a.astype(np.uint64)
    with np.errstate(over="ignore"):
        a = a + np.uint64(0x9E3779B97F4A7C15)
        a = (a ^ (a >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        a = (a ^ (a >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        a = a ^ (a >> np.uint64(31))
    return a


def det_matrix(n: int, d: int, seed: int) -> np.ndarray:
    """n x d float32 matrix with entries k/1024, k in [-2048, 2047] — exactly representable, so the
    same matrix is rebuilt bit-for-bit by tests/util.py on any platform."""
    idx = np.arange(n * d, dtype=np.uint64)
    with np.errstate(over="ignore"):
        h = hash_u64(idx * np.uint64(0x2545F4914F6CDD1D) + np.uint64(seed))
    k = (h % np.uint64(4096)).astype(np.int64) - 2048
    return (k.astype(np.float32) / F(1024.0)).reshape(n, d)


# ----------------------------------------------------------------------------- numpy restatement
def cos_dist_f32(q: np.ndarray, x: np.ndarray) -> np.float32:
    """1 - q.x / (|q||x|), sequential float32 accumulation (cumsum is sequential)."""
    dot = np.cumsum(q * x, dtype=np.float32)[-1]
    nq = np.cumsum(q * q, dtype=np.float32)[-1]
    nx = np.cumsum(x * x, dtype=np.float32)[-1]
    den = F(np.sqrt(nq)) * F(np.sqrt(nx))
    if not den > 0:
        return F(1.0)
    return F(F(1.0) - F(dot / den))


def cos_dist_f64(q: np.ndarray, x: np.ndarray) -> float:
    q = q.astype(np.float64)
    x = x.astype(np.float64)
    den = np.sqrt(q @ q) * np.sqrt(x @ x)
    return 1.0 if not den > 0 else float(1.0 - (q @ x) / den)


def vector_search(corpus, row_doc, q, k, dead=None, allow=None):
    hits = []
    for r in range(corpus.shape[0]):
        if dead is not None and dead[r]:
            continue
        doc = int(row_doc[r])
        if allow is not None and not (doc < len(allow) and allow[doc]):
            continue
        hits.append((cos_dist_f32(q, corpus[r]), r, doc))
    hits.sort(key=lambda h: (h[0], h[1]))          # selection: distance asc, row asc
    hits = hits[:k]
    hits.sort(key=lambda h: (h[0], h[2], h[1]))    # final order: distance asc, doc asc, row asc
    return hits


def rescale(score, is_e5):
    score = F(score)
    if not is_e5:
        return score
    mn, mx = F(0.7), F(1.0)
    delta = F(mx - mn)
    c = min(max(score, mn), mx)
    return F(F(c - mn) / delta)


def epilogue(hits, is_e5, min_sim):
    out = {}
    for dist, _row, doc in hits:
        sim = F(F(1.0) - dist)
        s = rescale(sim, is_e5)
        if s >= F(min_sim):
            out[doc] = F(out.get(doc, F(0.0)) + s)
    return out


def idf(n_docs, df):
    df = F(df)
    ratio = F(F(F(F(n_docs) - df) + F(0.5)) / F(df + F(0.5)))
    return F(np.log1p(ratio))


def ntf(tf, ln, avg, b):
    tf, ln, avg, b = F(tf), F(ln), F(avg), F(b)
    return F(tf / F(F(F(1.0) - b) + F(b * F(ln / avg))))


def bm25f(s, k, idf_):
    s, k, idf_ = F(s), F(k), F(idf_)
    return F(F(F(idf_ * F(k + F(1.0))) * s) / F(k + s))


def search_full_text(entries, n_tokens, n_docs, k=1.2, threshold=None):
    """entries: list of (token, [docs], [ntf]); returns {doc: score}."""
    scores, masks = {}, {}
    for t in range(n_tokens):
        contrib = {}
        for tok, docs, ntfs in entries:
            if tok != t:
                continue
            for d, v in zip(docs, ntfs):
                contrib.setdefault(int(d), []).append(F(v))
        df = max(len(contrib), 1)
        i = idf(n_docs, df)
        for d, vs in contrib.items():
            s = F(0.0)
            for v in vs:
                s = F(s + F(F(1.0) * v))
            if not np.isfinite(s) or s == 0 or abs(s) < np.finfo(np.float32).tiny:
                continue  # !is_normal
            ts = bm25f(s, k, i)
            if np.isnan(ts):
                continue
            scores[d] = F(scores.get(d, F(0.0)) + F(ts * F(1.0)))
            masks[d] = masks.get(d, 0) | (1 << (t % 32))  # u32 shift, amount masked (release-mode Rust)
    if threshold is not None:
        scores = {d: s for d, s in scores.items() if bin(masks[d]).count("1") >= threshold}
    return scores


def f32max(a, b):
    if np.isnan(a):
        return b
    if np.isnan(b):
        return a
    return a if a > b else b


def f32min(a, b):
    if np.isnan(a):
        return b
    if np.isnan(b):
        return a
    return a if a < b else b


def normalize_and_combine(vec: dict, ft: dict) -> dict:
    mx = F(0.0)
    for v in vec.values():
        mx = f32max(mx, v)
    m2 = F(0.0)
    for v in ft.values():
        m2 = f32max(m2, v)
    mx = f32max(mx, m2)
    mn = F(0.0)
    for v in vec.values():
        mn = f32min(mn, v)
    m2 = F(0.0)
    for v in ft.values():
        m2 = f32min(m2, v)
    mn = f32min(mn, m2)
    with np.errstate(invalid="ignore", divide="ignore"):
        out = {d: F(F(v - mn) / F(mx - mn)) for d, v in ft.items()}
        for d, v in vec.items():
            out[d] = F(out.get(d, F(0.0)) + F(F(v - mn) / F(mx - mn)))
    return out


def top_n(scores: dict, n: int):
    items = [(d, s) for d, s in scores.items() if not np.isnan(s)]
    items.sort(key=lambda t: (-float(t[1]), t[0]))
    return items[:n]


# ----------------------------------------------------------------------------- fixture writers
def jf(x):
    """float32 → JSON-safe (value as python float of the f32, plus raw bits for exactness)."""
    x = F(x)
    return {"v": None if np.isnan(x) else (float(x) if np.isfinite(x) else ("inf" if x > 0 else "-inf")),
            "bits": int(np.asarray(x).view(np.uint32))}


def make_bm25_kat():
    """bm25.rs:533-1043 as data."""
    k = F(1.2)
    cases = []
    # test_bm25f_scorer_basic  (bm25.rs:533-563)
    ratio = F(F(F(F(100.0) - F(10.0)) + F(0.5)) / F(F(10.0) + F(0.5)))
    eidf = F(np.log1p(ratio))
    e = F(F(F(eidf * F(F(1.2) + F(1.0))) * F(5.0)) / F(F(1.2) + F(5.0)))
    cases.append({"name": "test_bm25f_scorer_basic", "ref": "bm25.rs:533-563", "kind": "legacy_add",
                  "adds": [{"doc": 1, "tf": 5, "len": 100, "avglen": 100.0, "total_docs": 100.0, "df": 10,
                            "k": 1.2, "weight": 1.0, "b": 0.75, "boost": 1.0}],
                  "expect": {"1": float(e)}, "tol": 1e-6})
    # test_bm25f_scorer_boost (bm25.rs:565-616)
    cases.append({"name": "test_bm25f_scorer_boost", "ref": "bm25.rs:565-616", "kind": "legacy_add",
                  "adds": [{"doc": d, "tf": 5, "len": 100, "avglen": 100.0, "total_docs": 100.0, "df": 10,
                            "k": 1.2, "weight": 1.0, "b": 0.75, "boost": bo}
                           for d, bo in ((1, 1.0), (2, 2.0), (3, 0.5))],
                  "relations": [["gt", 2, 1], ["lt", 3, 1]]})
    # test_bm25f_field_weights (bm25.rs:618-668): two adds on doc1 > single-field expected
    cases.append({"name": "test_bm25f_field_weights", "ref": "bm25.rs:618-668", "kind": "legacy_add",
                  "adds": [{"doc": 1, "tf": 5, "len": 100, "avglen": 100.0, "total_docs": 100.0, "df": 10,
                            "k": 1.2, "weight": w, "b": 0.75, "boost": 1.0} for w in (2.0, 1.0)],
                  "relations": [["gt_value", 1, float(e)]], "n_docs": 1})
    # test_bm25f_field_normalization (bm25.rs:670-712)
    cases.append({"name": "test_bm25f_field_normalization", "ref": "bm25.rs:670-712", "kind": "legacy_add",
                  "adds": [{"doc": d, "tf": 5, "len": 200, "avglen": 100.0, "total_docs": 100.0, "df": 10,
                            "k": 1.2, "weight": 1.0, "b": b, "boost": 1.0} for d, b in ((1, 0.2), (2, 0.9))],
                  "relations": [["gt", 1, 2]]})
    # test_bm25f_boost_integration_single_field (bm25.rs:714-778): ratio > 1.05
    cases.append({"name": "test_bm25f_boost_integration_single_field", "ref": "bm25.rs:714-778",
                  "kind": "legacy_add",
                  "adds": [{"doc": d, "tf": 5, "len": 100, "avglen": 100.0, "total_docs": 100.0, "df": 10,
                            "k": 1.2, "weight": w, "b": 0.75, "boost": 1.0} for d, w in ((1, 1.0), (2, 2.0))],
                  "relations": [["ratio_gt", 2, 1, 1.05]]})
    # test_bm25f_boost_integration_multi_field (bm25.rs:780-866): title-only > content-only
    cases.append({"name": "test_bm25f_boost_integration_multi_field", "ref": "bm25.rs:780-866",
                  "kind": "legacy_add",
                  "adds": [{"doc": 1, "tf": 3, "len": 50, "avglen": 50.0, "total_docs": 100.0, "df": 10,
                            "k": 1.2, "weight": 3.0, "b": 0.75, "boost": 1.0},
                           {"doc": 2, "tf": 3, "len": 200, "avglen": 200.0, "total_docs": 100.0, "df": 10,
                            "k": 1.2, "weight": 1.0, "b": 0.75, "boost": 1.0}],
                  "relations": [["gt", 1, 2], ["gt_value", 1, 0.0]]})
    # test_bm25f_boost_values_comparison (bm25.rs:868-909)
    cases.append({"name": "test_bm25f_boost_values_comparison", "ref": "bm25.rs:868-909", "kind": "legacy_add",
                  "adds": [{"doc": i + 1, "tf": 5, "len": 100, "avglen": 100.0, "total_docs": 100.0, "df": 10,
                            "k": 1.2, "weight": w, "b": 0.75, "boost": 1.0}
                           for i, w in enumerate((0.5, 1.0, 1.5, 2.0, 3.0))],
                  "relations": [["gt", 2, 1], ["gt", 3, 2], ["gt", 4, 3], ["gt", 5, 4],
                                ["ratio_gt", 4, 2, 1.0], ["ratio_lt", 4, 2, 1.5]]})
    # test_canonical_bm25f_single_term_two_fields (bm25.rs:911-983)
    t_ntf = F(F(2.0) / F(F(F(1.0) - F(0.75)) + F(F(0.75) * F(F(10.0) / F(8.0)))))
    c_ntf = F(F(1.0) / F(F(F(1.0) - F(0.75)) + F(F(0.75) * F(F(200.0) / F(150.0)))))
    agg = F(F(F(2.0) * t_ntf) + F(F(1.0) * c_ntf))
    e2 = F(F(F(eidf * F(k + F(1.0))) * agg) / F(k + agg))
    cases.append({"name": "test_canonical_bm25f_single_term_two_fields", "ref": "bm25.rs:911-983",
                  "kind": "add_field",
                  "fields": [{"doc": 1, "tf": 2, "len": 10, "avglen": 8.0, "weight": 2.0, "b": 0.75},
                             {"doc": 1, "tf": 1, "len": 200, "avglen": 150.0, "weight": 1.0, "b": 0.75}],
                  "finalize": {"df": 10, "total_docs": 100.0, "k": 1.2},
                  "expect": {"1": float(e2)}, "tol": 1e-5})
    # test_canonical_bm25f_vs_sum_of_per_field_bm25 (bm25.rs:985-1043)
    f1 = F(F(3.0) / F(F(F(1.0) - F(0.75)) + F(F(0.75) * F(F(50.0) / F(40.0)))))
    f2 = F(F(2.0) / F(F(F(1.0) - F(0.75)) + F(F(0.75) * F(F(100.0) / F(80.0)))))
    i1 = F(F(F(F(F(2.0) * eidf) * F(k + F(1.0))) * f1) / F(k + f1))
    i2 = F(F(F(F(F(1.0) * eidf) * F(k + F(1.0))) * f2) / F(k + f2))
    cases.append({"name": "test_canonical_bm25f_vs_sum_of_per_field_bm25", "ref": "bm25.rs:985-1043",
                  "kind": "add_field",
                  "fields": [{"doc": 1, "tf": 3, "len": 50, "avglen": 40.0, "weight": 2.0, "b": 0.75},
                             {"doc": 1, "tf": 2, "len": 100, "avglen": 80.0, "weight": 1.0, "b": 0.75}],
                  "finalize": {"df": 10, "total_docs": 100.0, "k": 1.2},
                  "relations": [["le_value", 1, float(F(i1 + i2)) + 1e-6], ["gt_value", 1, 0.0]]})
    (HERE / "bm25_kat.json").write_text(json.dumps({"cases": cases}, indent=1))


def make_cosine():
    out = {}
    for d in (384, 768):
        n, nq, k = 4096, 8, 100
        corpus = det_matrix(n, d, seed=1000 + d)
        queries = det_matrix(nq, d, seed=2000 + d)
        row_doc = np.arange(n, dtype=np.uint64)
        ids = np.zeros((nq, k), dtype=np.uint64)
        d32 = np.zeros((nq, k), dtype=np.float32)
        d64 = np.zeros((nq, k), dtype=np.float64)
        for qi in range(nq):
            hits = vector_search(corpus, row_doc, queries[qi], k)
            for j, (dist, row, doc) in enumerate(hits):
                ids[qi, j] = doc
                d32[qi, j] = dist
                d64[qi, j] = cos_dist_f64(queries[qi], corpus[row])
        out[f"ids_{d}"] = ids
        out[f"dist32_{d}"] = d32
        out[f"dist64_{d}"] = d64
    np.savez_compressed(HERE / "cosine_small.npz", **out)


def make_cosine_ties_multirow():
    # ties: 64 distinct rows, each duplicated 4x under different doc ids (shuffled deterministically)
    d, base_n = 384, 64
    base = det_matrix(base_n, d, seed=77)
    order = np.argsort(hash_u64(np.arange(base_n * 4, dtype=np.uint64) + np.uint64(5)), kind="stable")
    corpus = np.concatenate([base] * 4, axis=0)[order]
    row_doc = (np.arange(base_n * 4, dtype=np.uint64) * np.uint64(3) + np.uint64(7))[
        np.argsort(hash_u64(np.arange(base_n * 4, dtype=np.uint64) + np.uint64(11)), kind="stable")]
    q = det_matrix(1, d, seed=78)[0]
    res = {}
    for k in (1, 2, 3, 5, 10, 17, 100, 256):
        hits = vector_search(corpus, row_doc, q, k)
        res[str(k)] = {"ids": [h[2] for h in hits], "rows": [h[1] for h in hits],
                       "dist_bits": [int(np.asarray(h[0]).view(np.uint32)) for h in hits]}
    (HERE / "cosine_ties.json").write_text(json.dumps(res))

    # multirow: 300 docs with 1..5 rows each; some docs deleted, a filter; E5 + non-E5 epilogue
    d = 384
    n_docs = 300
    rows_per = (hash_u64(np.arange(n_docs, dtype=np.uint64) + np.uint64(123)) % np.uint64(5)).astype(int) + 1
    row_doc = np.repeat(np.arange(n_docs, dtype=np.uint64) + np.uint64(1000), rows_per)
    n = int(row_doc.shape[0])
    corpus = det_matrix(n, d, seed=99)
    q = det_matrix(1, d, seed=98)[0]
    # plant near-duplicates of q so that similarities cross the 0.7 cut-off
    for j, r in enumerate(range(0, n, 37)):
        alpha = F(0.5 + 0.05 * j)
        corpus[r] = (alpha * q + (F(1.0) - alpha) * corpus[r]).astype(np.float32)
    dead_docs = {1003, 1050, 1100}
    dead = np.array([int(x) in dead_docs for x in row_doc], dtype=np.uint8)
    allow = np.zeros(1400, dtype=bool)
    allow[1000:1300:2] = True
    allow[1001] = True
    res = {"n_rows": n}
    for name, kw in (("plain", {}), ("dead", {"dead": dead}), ("filter", {"allow": allow}),
                     ("dead_filter", {"dead": dead, "allow": allow})):
        for k in (5, 10, 50):
            hits = vector_search(corpus, row_doc, q, k, **kw)
            entry = {"ids": [h[2] for h in hits], "rows": [h[1] for h in hits],
                     "dist": [float(h[0]) for h in hits]}
            for is_e5 in (0, 1):
                for ms in (0.0, 0.7):
                    m = epilogue(hits, is_e5, ms)
                    entry[f"map_e5{is_e5}_min{ms}"] = {str(dk): float(v) for dk, v in sorted(m.items())}
            res[f"{name}_k{k}"] = entry
    (HERE / "cosine_multirow.json").write_text(json.dumps(res))


def zipf_corpus(n_docs, vocab, n_fields, seed):
    """Deterministic synthetic postings: per field, term -> sorted (doc, tf), plus field lengths."""
    fields = []
    for f in range(n_fields):
        lens = (hash_u64(np.arange(n_docs, dtype=np.uint64) * np.uint64(7) + np.uint64(seed + f)) % np.uint64(40)
                ).astype(np.int64) + 4
        postings = {}
        for dix in range(n_docs):
            L = int(lens[dix])
            h = hash_u64(np.arange(L, dtype=np.uint64) + np.uint64(dix * 1000003 + seed * 31 + f * 17))
            u = (h >> np.uint64(11)).astype(np.float64) / float(1 << 53)
            terms = np.minimum((vocab ** u).astype(np.int64) - 1, vocab - 1)  # log-uniform ≈ Zipf(1)
            for t, c in zip(*np.unique(terms, return_counts=True)):
                postings.setdefault(int(t), []).append((dix, int(c)))
        fields.append({"lens": lens, "postings": postings, "avg": float(np.float32(lens.mean()))})
    return fields


def make_bm25_synth():
    n_docs, vocab, n_fields = 2000, 500, 3
    fields = zipf_corpus(n_docs, vocab, n_fields, seed=4242)
    doc_ids = (np.arange(n_docs, dtype=np.uint64) * np.uint64(2) + np.uint64(10))  # non-dense ids
    boosts = [1.0, 2.0, 0.5]
    b = 0.75
    queries = []
    hq = hash_u64(np.arange(64, dtype=np.uint64) + np.uint64(9001))
    for qi in range(6):
        n_tok = [1, 2, 3, 5, 8, 12][qi]
        toks = [int(hq[(qi * 12 + j) % 64] % np.uint64(vocab // 2)) for j in range(n_tok)]
        queries.append(toks)
    allow = (hash_u64(doc_ids + np.uint64(5)) % np.uint64(3)) != 0  # ~2/3 of docs allowed
    cases = []
    for qi, toks in enumerate(queries):
        for threshold_frac in (None, 0.5, 1.0):
            for use_filter in (False, True):
                entries = []
                for ti, term in enumerate(toks):
                    for f in range(n_fields):
                        pl = fields[f]["postings"].get(term, [])
                        docs, ntfs = [], []
                        for dix, tf in pl:
                            if use_filter and not allow[dix]:
                                continue
                            v = F(F(boosts[f]) * ntf(tf, int(fields[f]["lens"][dix]), F(fields[f]["avg"]), b))
                            docs.append(int(doc_ids[dix]))
                            ntfs.append(v)
                        entries.append((ti, docs, ntfs))
                thr = None if threshold_frac is None else int(np.floor(F(len(toks)) * F(threshold_frac)))
                scores = search_full_text(entries, len(toks), float(n_docs), 1.2, thr)
                top = top_n(scores, 20)
                cases.append({"query": qi, "terms": toks, "threshold": thr, "filter": use_filter,
                              "count": len(scores),
                              "top_ids": [int(d) for d, _ in top],
                              "top_scores": [float(s) for _, s in top],
                              "checksum": float(np.sum(np.array([float(s) for s in scores.values()],
                                                                dtype=np.float64)))})
    meta = {"n_docs": n_docs, "vocab": vocab, "n_fields": n_fields, "seed": 4242, "boosts": boosts, "b": b,
            "doc_id_mul": 2, "doc_id_add": 10, "allow_rule": "hash_u64(doc_id + 5) % 3 != 0", "cases": cases}
    (HERE / "bm25_synth.json").write_text(json.dumps(meta))


def make_hybrid_omc():
    cases = []

    def run(name, vec, ft, omc=None, k=10):
        comb = normalize_and_combine({d: F(s) for d, s in vec.items()}, {d: F(s) for d, s in ft.items()})
        if omc:
            comb = {d: (F(s * F(omc[d])) if d in omc else s) for d, s in comb.items()}
        top = top_n(comb, k)
        cases.append({"name": name, "vec": {str(d): s for d, s in vec.items()},
                      "ft": {str(d): s for d, s in ft.items()}, "omc": {str(d): s for d, s in (omc or {}).items()},
                      "k": k, "count": len(comb),
                      "combined": {str(d): jf(s) for d, s in sorted(comb.items())},
                      "top_ids": [int(d) for d, _ in top], "top_scores": [jf(s) for _, s in top]})

    run("all_positive_min_is_zero", {1: 0.9, 2: 0.8, 7: 0.75}, {1: 3.2, 3: 1.1, 4: 0.4, 7: 2.0})
    run("negative_vector_scores", {1: -0.2, 2: 0.5, 9: -0.6}, {1: 2.0, 3: 1.0, 9: 0.25})
    run("max_equals_min_gives_nan", {}, {})
    run("all_zero_scores_nan_dropped", {1: 0.0, 2: 0.0}, {3: 0.0})
    run("vector_only", {5: 0.71, 6: 0.93}, {})
    run("fulltext_only", {}, {5: 1.5, 6: 0.5, 8: 4.0})
    run("ties_after_combine", {1: 0.5, 2: 0.5}, {1: 1.0, 2: 1.0, 3: 2.0, 4: 2.0}, k=3)
    # OMC ratios mirroring src/tests/omc_test.rs:485-553 (0.25 / 0.5 / 2 / 5 / 10)
    run("omc_ratios", {1: 0.9, 2: 0.8}, {1: 1.0, 2: 1.0, 3: 1.0, 4: 1.0, 5: 1.0, 6: 1.0},
        omc={1: 0.25, 2: 0.5, 3: 2.0, 4: 5.0, 5: 10.0})
    (HERE / "hybrid_kat.json").write_text(json.dumps({"cases": cases}, indent=1))


def make_fulltext_ordinal():
    """Ordinal replica of src/tests/fulltext_search.rs:192-251: 100 docs, doc i holds "text " x (i+1);
    query "text" must rank 99, 98, 97, … (longer doc = higher tf; with b = 0.75 the tf growth wins),
    limit 10, count 100."""
    n = 100
    lens = np.arange(1, n + 1)
    avg = F(lens.mean())
    docs = list(range(n))
    ntfs = [ntf(int(L), int(L), avg, 0.75) for L in lens]
    scores = search_full_text([(0, docs, ntfs)], 1, float(n))
    top = top_n(scores, 10)
    (HERE / "fulltext_ordinal.json").write_text(json.dumps(
        {"n": n, "avg": float(avg), "count": len(scores), "top_ids": [int(d) for d, _ in top],
         "top_scores": [float(s) for _, s in top]}))


if __name__ == "__main__":
    make_bm25_kat()
    make_cosine()
    make_cosine_ties_multirow()
    make_bm25_synth()
    make_hybrid_omc()
    make_fulltext_ordinal()
    print("golden fixtures written to", HERE)
