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
        self.bool_fields, self.number_fields, self.string_filter_fields, self.date_fields = {}, {}, {}, {}
        self.uncommitted_deleted_documents = set()

    def delete_documents(self, doc_ids):
        """IndexWriteOperation::DeleteDocuments (index/mod.rs:1346-1424) on the host-side state (the mirror's Index does the same
        and keeps the device in step)."""
        for d in doc_ids:
            if d not in self.document_ids:
                continue
            self.document_ids.discard(d)
            for sf in self.string_fields.values():
                sf.delete(d)
            for store in (self.bool_fields, self.number_fields, self.string_filter_fields, self.date_fields):
                for vals in store.values():
                    vals.pop(d, None)
            self.omc.pop(d, None)

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
        v = doc.get("_omc")
        # write/index/mod.rs:451-458: only a positive NUMBER is a multiplier (as_f64, > 0); anything else is ignored
        if isinstance(v, (int, float)) and not isinstance(v, bool) and float(v) > 0.0:
            idx.omc[n] = float(v)
    for gone in case.get("delete_ids", []):  # delete_documents: out of the document set, the fields and the OMC map
        n = ids.get(gone)
        if n is None:
            continue  # (an id the index never held: src/tests/delete_doc.rs:72-120)
        idx.document_ids.discard(n)
        for sf in idx.string_fields.values():
            sf.delete(n)
        idx.omc.pop(n, None)
    return ids


def levenshtein(a: bytes, b: bytes) -> int:
    prev = list(range(len(b) + 1))
    for i, ca in enumerate(a, start=1):
        cur = [i]
        for j, cb in enumerate(b, start=1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (ca != cb)))
        prev = cur
    return prev[-1]


DEFAULT_EXACT_MATCH_BOOST = 1.5  # = oramacore_amd.token_score.DEFAULT_EXACT_MATCH_BOOST (asserted in tests/test_reference_cases.py)


def oracle_search(idx, p, fields, exact_match_boost=DEFAULT_EXACT_MATCH_BOOST, allowed=None):
    """One search of a case through the CPU restatement: (docs, scores) of the whole map after OMC.  `allowed`: the set of
    documents that pass the filter (collect_contributions_with_filter: the others never reach the scorer), None = no filter."""
    fmap = {name: fi for fi, name in enumerate(fields)}
    exact = p.get("exact", False)
    tokens = tokens_of(p["term"], exact)
    props = sorted(idx.string_fields) if "properties" not in p else sorted(fmap[n] for n in p["properties"])
    boost = {fmap[n]: float(v) for n, v in p.get("boost", {}).items()}
    tol = int(p.get("tolerance") or 0)
    entries = []
    for ti, tok in enumerate(tokens):
        for fid in props:
            sf = idx.string_fields[fid]
            for term in sorted(sf.postings):
                # the dictionary step: the term itself (exact), or every term the token prefixes plus, with `tolerance`, every
                # term within that Levenshtein distance (bytes) — src/tests/fulltext_search.rs:603-753, 956-1018
                if not (term == tok if exact else (term.startswith(tok) or (tol and levenshtein(term.encode(), tok.encode()) <= tol))):
                    continue
                bo = F(boost.get(fid, 1.0))
                if term == tok and exact_match_boost != 1.0:
                    bo = F(bo * F(exact_match_boost))
                pl = sorted((d, tf) for d, tf in sf.postings[term].items() if allowed is None or d in allowed)
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
    for a, b in exp.get("rank_before", []):
        assert got.index(a) < got.index(b), (a, b, got)
    for i in exp.get("contains_ids", []):
        assert i in got, (i, got)
    for i in range(exp.get("strictly_decreasing_scores", 1) - 1):
        assert hits[i][1] > hits[i + 1][1], (i, hits[i], hits[i + 1])
    if exp.get("positive_scores"):
        assert all(h[1] > 0.0 for h in hits), hits
    if exp.get("scores_bit_equal"):
        assert len({np.float32(h[1]).view(np.uint32).item() for h in hits}) == 1, hits


def check_case(case, search):
    """Run every search of `case` through `search(spec, exact_match_boost) -> (hits [(doc, score)], count, ids)` and assert
    everything the reference's test asserts (+ the relations marked `own`)."""
    assert "host_params" not in case, "round 6: every case runs under the declared defaults — no per-case host parameter"
    emb = DEFAULT_EXACT_MATCH_BOOST
    saved = {}
    for spec in case["searches"]:
        hits, count, ids = search(spec, emb)
        check_expect(hits, count, ids, spec["expect"])
        if "save_top_score_as" in spec:
            saved[spec["save_top_score_as"]] = float(hits[0][1])
        for name, doc_id in spec.get("save_scores", {}).items():  # the score of ONE document among the hits
            saved[name] = float(dict(hits)[ids[doc_id]])
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
        for e in (emb, 1.0):  # the declared default and the factor-free variant
            if e != emb:
                for spec in case["searches"]:
                    if "save_top_score_as" in spec:
                        saved[spec["save_top_score_as"]] = float(search(spec, e)[0][0][1])
            want = {n: n * (cf["k"] + e) / (cf["k"] + n * e) for n in (2, 5)}
            assert abs(saved["s2"] / saved["s1"] - want[2]) <= cf["tol"] and abs(saved["s5"] / saved["s1"] - want[5]) <= cf["tol"], (e, want)
    alt = case.get("with_exact_match_boost_1")
    if alt:  # the declared default (1.0): same documents, same order by id, but the scores tie — the factor is what the reference pins
        hits, count, ids = search(alt, 1.0)
        check_expect(hits, count, ids, alt["expect"])


# ---------------------------------------------------------------- facet cases (tests/golden/reference_facet_cases.json)
def fill_filters(idx, case, ids):
    """The case's filter-field values (`values[i]` = document i in insert order) into idx.bool_fields / number_fields /
    string_filter_fields ({DocumentId: value}) — HostIndex or the mirror's Index (before its commit)."""
    order = [ids[d["id"]] for d in documents_of(case)]
    for name, spec in case.get("filter_fields", {}).items():
        store = {"bool": idx.bool_fields, "number": idx.number_fields, "string": idx.string_filter_fields}[spec["kind"]]
        assert len(spec["values"]) == len(order), name
        store[name] = dict(zip(order, spec["values"]))


def facet_definitions(case):
    """JSON -> what token_score.facets_and_groups takes: "bool" | "string" | [(from, to), ...]."""
    return {n: (d if isinstance(d, str) else [tuple(r) for r in d["ranges"]]) for n, d in case["facets"].items()}


def where_docs(idx, where):
    """Documents that satisfy the case's `where` (field == value on a string filter field), or None."""
    if not where:
        return None
    keep = None
    for name, value in where.items():
        hit = {d for d, v in idx.string_filter_fields[name].items() if v == value}
        keep = hit if keep is None else keep & hit
    return keep


def oracle_facets(idx, map_docs, definitions):
    """facet.rs:150-206 over the score map `map_docs` through the oracle's counting functions: {name: {"count", "values"}};
    a field this index does not hold is skipped (facet.rs:159-163)."""
    out = {}
    live = idx.document_ids
    for name, definition in definitions.items():
        if isinstance(definition, list):
            if name not in idx.number_fields:
                continue
            docs = [d for d in idx.number_fields[name] if d in live]
            nums = [float(idx.number_fields[name][d]) for d in docs]
            counts = orc.facet_count_ranges(map_docs, docs, nums, definition)
            values = {f"{a}-{b}": int(c) for (a, b), c in zip(definition, counts)}
        else:
            store = idx.bool_fields if definition == "bool" else idx.string_filter_fields
            if name not in store:
                continue
            keys = [True, False] if definition == "bool" else sorted({v for d, v in store[name].items() if d in live})
            buckets = [sorted(d for d, v in store[name].items() if (v is k if definition == "bool" else v == k) and d in live) for k in keys]
            off = np.concatenate([[0], np.cumsum([len(b) for b in buckets])]).astype(np.uint64)
            flat = np.concatenate([np.asarray(b, dtype=np.uint64) for b in buckets]) if sum(map(len, buckets)) else np.zeros(0, dtype=np.uint64)
            counts = orc.facet_count_buckets(map_docs, off, flat)
            labels = [("true" if k else "false") for k in keys] if definition == "bool" else keys
            values = {str(l): int(c) for l, c in zip(labels, counts)}
        out[name] = {"count": len(values), "values": values}
    return out


def add_facets(total, part):
    """Facet results of one more index into the collection's (values added, `count` = distinct values: search.rs:398-420)."""
    for name, r in part.items():
        t = total.setdefault(name, {"count": 0, "values": {}})
        for k, v in r["values"].items():
            t["values"][k] = t["values"].get(k, 0) + v
        t["count"] = len(t["values"])
    return total


def check_facets(case, got):
    if case["expect"].get("error") == "FacetFieldNotFound":  # search.rs:452-463, over the collection's summed facets
        from oramacore_amd.token_score import FacetFieldNotFound, check_facet_results
        try:
            check_facet_results(case["facets"], got)
        except FacetFieldNotFound as e:
            assert list(e.args[0]) == case["expect"]["missing"]
            return
        raise AssertionError("expected FacetFieldNotFound")
    from oramacore_amd.token_score import check_facet_results
    check_facet_results(case["facets"], got)
    for name, exp in case["expect"].items():
        assert name in got, (case["name"], name)
        assert got[name]["values"] == exp["values"], (case["name"], name, got[name]["values"])
        if "count" in exp:
            assert got[name]["count"] == exp["count"], (case["name"], name)


# ---------------------------------------------------------------- group cases (tests/golden/reference_group_cases.json)
def group_variants(idx, name):
    """calculate_group_for_field (group.rs:181-270): value -> documents, over whichever filter store holds the field."""
    for store in (idx.bool_fields, idx.number_fields, idx.string_filter_fields):
        if name in store:
            out = {}
            for d, v in store[name].items():
                if d in idx.document_ids:
                    out.setdefault(v, set()).add(d)
            return out
    return None


def oracle_groups(idx, map_docs, map_scores, properties, max_results):
    """group.rs:113-160 + sort.rs:203-213 through the oracle: {tuple(values): [(doc, score)]} — one group per combination of
    the properties' values, its documents = the intersection of the values' document sets, the best `max_results` of them that
    are in the score map by (score desc, DocumentId asc)."""
    import itertools

    variants = [group_variants(idx, p) for p in properties]
    if any(v is None for v in variants):
        return {}
    combos, buckets = [], []
    for combo in itertools.product(*[sorted(v, key=str) for v in variants]):
        combos.append(combo)
        buckets.append(sorted(set.intersection(*[variants[i][val] for i, val in enumerate(combo)])))
    off = np.concatenate([[0], np.cumsum([len(b) for b in buckets])]).astype(np.uint64)
    flat = np.concatenate([np.asarray(b, dtype=np.uint64) for b in buckets]) if sum(map(len, buckets)) else np.zeros(0, dtype=np.uint64)
    if max_results == 0:
        return {c: [] for c in combos}
    g_ids, g_sc, g_n = orc.group_top(np.asarray(map_docs, dtype=np.uint64), np.asarray(map_scores, dtype=F), off, flat, max_results)
    return {c: list(zip(i_[: int(n_)].tolist(), s_[: int(n_)].tolist())) for c, i_, s_, n_ in zip(combos, g_ids, g_sc, g_n)}


def check_groups(spec, groups, ids, hits=None, count=None):
    """`groups`: {tuple(values): [(doc, score)]}; the reference lists only groups... of every combination (empty ones included)."""
    exp = spec["expect"]
    if "n_groups" in exp:
        assert len(groups) == exp["n_groups"], sorted(groups, key=str)
    if "count" in exp and count is not None:
        assert count == exp["count"]
    back = {v: k for k, v in ids.items()}
    norm = lambda v: tuple(float(x) if isinstance(x, (int, float)) and not isinstance(x, bool) else x for x in v)
    if "group_values" in exp:
        assert {norm(k) for k in groups} == {norm(v) for v in exp["group_values"]}, sorted(groups, key=str)
    if "max_len" in exp:
        assert all(len(g) <= exp["max_len"] for g in groups.values())
    for name, want in exp.get("groups", {}).items():
        assert [back[d] for d, _ in groups[(name,)]] == want, (name, groups[(name,)])
    for name, n in exp.get("group_lens", {}).items():
        assert len(groups[(name,)]) == n, (name, groups[(name,)])
    if "group_lens" in exp:
        assert len(groups) == len(exp["group_lens"])
    if "hit_ids" in exp and hits is not None:
        assert [back[h[0]] for h in hits] == exp["hit_ids"]
    for g in groups.values():  # own: score-ordered, ties by DocumentId
        assert all((a[1], -a[0]) >= (b[1], -b[0]) for a, b in zip(g, g[1:])), g
