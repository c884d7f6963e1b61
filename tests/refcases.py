"""Shared by tests/test_reference_cases.py (CPU: the oracle against the reference's cases) and tests/test_reference_cases_gpu.py
(the HIP path against the same cases and against the oracle): tests/golden/reference_cases.json read into host-side postings."""
import numpy as np

from oracle import oracle as orc
from oramacore_amd.token_score import SimpleTokenizer, StringFieldStorage

F = np.float32


class HostIndex:
    """What the oracle needs of an index, without a GPU: string fields, OMC map, document count."""

    def __init__(self):
        self.string_fields, self.omc, self.document_ids = {}, {}, set()

    @property
    def document_count(self):
        return len(self.document_ids)


def tokens_of(term, exact):
    toks = SimpleTokenizer().tokenize_and_stem(term)
    out = [t for t, _ in toks] if exact else [x for t, s in toks for x in ((t,) if s is None else (t, s))]
    return out or [""]


def documents_of(case):
    if "documents" in case:
        return case["documents"]
    g = case["generate_documents"]
    assert g["repeat"] == "i + 1"
    return [{"id": g["id"].format(i=i), g["field"]: (g["word"] + " ") * (i + 1)} for i in range(g["count"])]


def fill(idx, case):
    """Insert the case's documents into `idx` (HostIndex or the mirror's Index); returns {string id: DocumentId}."""
    fields = case["fields"]
    for fi in range(len(fields)):
        idx.string_fields[fi] = StringFieldStorage()
    ids = {}
    for n, doc in enumerate(documents_of(case), start=1):  # DocumentIds are handed out sequentially at insert
        ids[doc["id"]] = n
        idx.document_ids.add(n)
        for fi, name in enumerate(fields):
            if name in doc:
                idx.string_fields[fi].insert(n, doc[name])
        if "_omc" in doc:
            idx.omc[n] = float(doc["_omc"])
    return ids


def oracle_search(idx, p, fields, exact_match_boost=1.0):
    """One search of a case through the CPU restatement: (docs, scores) of the whole map after OMC."""
    fmap = {name: fi for fi, name in enumerate(fields)}
    exact = p.get("exact", False)
    tokens = tokens_of(p["term"], exact)
    props = sorted(idx.string_fields) if "properties" not in p else sorted(fmap[n] for n in p["properties"])
    boost = {fmap[n]: float(v) for n, v in p.get("boost", {}).items()}
    entries = []
    for ti, tok in enumerate(tokens):
        for fid in props:
            sf = idx.string_fields[fid]
            for term in sorted(sf.postings):
                if not (term == tok if exact else term.startswith(tok)):
                    continue
                bo = F(boost.get(fid, 1.0))
                if term == tok and exact_match_boost != 1.0:
                    bo = F(bo * F(exact_match_boost))
                pl = sorted(sf.postings[term].items())
                entries.append((ti, [d for d, _ in pl],
                                [F(bo * orc.bm25f_normalized_tf(tf, sf.field_len[d], sf.avg_field_length(), 0.75)) for d, tf in pl]))
    thr = None if p.get("threshold") is None else int(np.floor(F(len(tokens)) * F(p["threshold"])))
    docs, scores = orc.search_full_text(entries, len(tokens), float(idx.document_count), 1.2, thr)
    if idx.omc:
        scores = orc.apply_omc(docs, scores, list(idx.omc), list(idx.omc.values()))
    return docs, scores


def check_expect(hits, count, ids, exp):
    back = {v: k for k, v in ids.items()}
    got = [back[h[0]] for h in hits]
    if "count" in exp:
        assert count == exp["count"]
    if "min_count" in exp:
        assert count >= exp["min_count"]
    if "n_hits" in exp:
        assert len(hits) == exp["n_hits"]
    if "ids" in exp:
        assert got == exp["ids"]
    if "ids_prefix" in exp:
        assert got[: len(exp["ids_prefix"])] == exp["ids_prefix"]
    if "first_id" in exp:
        assert got[0] == exp["first_id"]
    for i in range(exp.get("strictly_decreasing_scores", 1) - 1):
        assert hits[i][1] > hits[i + 1][1], (i, hits[i], hits[i + 1])
    if exp.get("scores_bit_equal"):
        assert len({np.float32(h[1]).view(np.uint32).item() for h in hits}) == 1, hits


def check_case(case, search):
    """Run every search of `case` through `search(spec, exact_match_boost) -> (hits [(doc, score)], count, ids)` and assert
    everything the reference's test asserts (+ the relations marked `own`)."""
    emb = case.get("host_params", {}).get("exact_match_boost", 1.0)
    saved = {}
    for spec in case["searches"]:
        hits, count, ids = search(spec, emb)
        check_expect(hits, count, ids, spec["expect"])
        if "save_top_score_as" in spec:
            saved[spec["save_top_score_as"]] = float(hits[0][1])
        by_id = {k: dict(hits).get(v) for k, v in ids.items()}
        for r in spec.get("score_ratios", []):
            assert abs(by_id[r["a"]] - by_id[r["b"]] * r["ratio"]) <= r["abs_tol"], (r, by_id)
    for r in case.get("relations", []):
        a, b = saved[r["a"]], saved[r["b"]]
        if r["kind"] == "greater":
            assert a > b, r
        elif r["kind"] == "ratio_greater":
            assert a / b > r["bound"], (r, a / b)
        elif r["kind"] == "ratio_less":
            assert a / b < r["bound"], (r, a / b)
    cf = case.get("closed_form")
    if cf:
        assert abs(saved["s2"] / saved["s1"] - cf["ratio_2x"]) <= cf["tol"] and abs(saved["s5"] / saved["s1"] - cf["ratio_5x"]) <= cf["tol"]
    alt = case.get("with_exact_match_boost_1")
    if alt:  # the declared default (1.0): same documents, same order by id, but the scores tie — the factor is what the reference pins
        hits, count, ids = search(alt, 1.0)
        check_expect(hits, count, ids, alt["expect"])
