"""Shared by tests/test_reference_filter_cases.py (CPU: oracle + the filter mirror) and tests/test_reference_filter_cases_gpu.py
(the HIP path): tests/golden/reference_filter_cases.json — the reference's filter / multi-index / update tests as data — turned
into a small collection model: indexes with collection-unique DocumentIds, inserts that replace, merge-updates, deletes, commits."""
import numpy as np

import refcases
from oracle import oracle as orc
from oramacore_amd.filter import FilterContext, FilterFieldNotFound, check_filter_fields, parse_date
from oramacore_amd.token_score import StringFieldStorage


def gen_value(spec, i):
    if "list" in spec:
        return [gen_value(spec["list"], i)]
    if "repeat" in spec:
        return spec["repeat"] * (i + spec["times_plus"])
    if "index" in spec:
        return i
    if "format_bool" in spec:
        m, r = spec["mod_eq"]
        return spec["format_bool"].format(b=str(i % m == r).lower())
    if "mod_eq" in spec:
        m, r = spec["mod_eq"]
        return i % m == r
    raise ValueError(spec)


def documents_of(index_spec):
    if "documents" in index_spec:
        return [dict(d) for d in index_spec["documents"]]
    g = index_spec["generate"]
    return [{"id": g["id"].format(i=i), **{k: gen_value(v, i) for k, v in g["values"].items()}} for i in range(g["count"])]


def _vals(v):
    return v if isinstance(v, list) else [v]


def _is_string_value(v):
    return all(isinstance(x, str) for x in _vals(v)) and len(_vals(v)) > 0


class UnknownIndex(Exception):
    """ReadError::UnknownIndex: a request names an index the collection does not hold (src/tests/fulltext_search.rs:459-473)."""


class Collection:
    """The case's collection.  `make_index()` -> HostIndex (CPU) or the mirror's Index (GPU); `on_insert(idx, [doc ids])`,
    `on_commit(idx)` let the GPU harness keep the device in step (delta lists / rebuild)."""

    def __init__(self, case, make_index, on_insert=None, on_commit=None):
        self.case = case
        self.on_insert, self.on_commit = on_insert or (lambda idx, ids: None), on_commit or (lambda idx: None)
        self.next_doc, self.auto = 1, 0
        self.indexes, self.ids, self.docs, self.fields = [], [], [], []
        self.deleted_indexes = set()
        later = {}
        for st in case["steps"]:
            if st.get("op") in ("insert", "update_merge"):
                later.setdefault(st["index"], []).extend(st["documents"])
        for ii, spec in enumerate(case["indexes"]):
            idx = make_index()
            docs = documents_of(spec)
            # the scored string fields of the index: every name that holds a string anywhere in the case, in order of appearance
            # (a field that first appears after a commit would need one more: the cases insert no such field)
            names = []
            for d in docs + later.get(ii, []):
                for k, v in d.items():
                    if k != "id" and _is_string_value(v) and k not in names:
                        names.append(k)
            for fi in range(len(names)):
                idx.string_fields[fi] = StringFieldStorage()
            if hasattr(idx, "path_to_field_id_map"):  # (the mirror's Index resolves property NAMES itself)
                idx.path_to_field_id_map.update({n: (fi, "string") for fi, n in enumerate(names)})
            self.indexes.append(idx)
            self.ids.append({})
            self.docs.append({})
            self.fields.append(names)
            new = [self._insert_host(ii, d) for d in docs]
            self.on_commit(idx)  # the initial batch is committed: the postings are resident
            del new

    # ---- host-side state (what both harnesses share)
    def _insert_host(self, ii, doc):
        idx, ids = self.indexes[ii], self.ids[ii]
        sid = doc.get("id")
        if sid is None:
            self.auto += 1
            sid = f"auto-{self.auto}"
        if sid in ids:  # replace on insert (src/tests/replace_doc_on_insert.rs): the old document goes first
            idx.delete_documents([ids[sid]])
        n = self.next_doc
        self.next_doc += 1
        ids[sid] = n
        self.docs[ii][sid] = dict(doc)
        idx.document_ids.add(n)
        omc = doc.get("_omc")  # write/index/mod.rs:451-458: only positive numbers are multipliers
        if isinstance(omc, (int, float)) and not isinstance(omc, bool) and omc > 0:
            idx.omc[n] = float(omc)
        for k, v in doc.items():
            if k in ("id", "_omc"):
                continue
            vals = _vals(v)
            if _is_string_value(v):
                idx.string_fields[self.fields[ii].index(k)].insert(n, " ".join(vals))
                # the FILTER field of a string value: a date field when the value parses as a date, else a string filter field
                # (write/index/mod.rs:811-818; the field's type is that of its first value)
                dates = [parse_date(x) for x in vals]
                if k in idx.date_fields or (k not in idx.string_filter_fields and all(t is not None for t in dates)):
                    if all(t is not None for t in dates):
                        idx.date_fields.setdefault(k, {})[n] = dates if isinstance(v, list) else dates[0]
                else:
                    idx.string_filter_fields.setdefault(k, {})[n] = v
            elif all(isinstance(x, bool) for x in vals):
                idx.bool_fields.setdefault(k, {})[n] = v
            elif all(isinstance(x, (int, float)) and not isinstance(x, bool) for x in vals):
                idx.number_fields.setdefault(k, {})[n] = v
        return n

    def apply(self, step):
        op = step["op"]
        if op == "commit":
            for idx in self.indexes:
                self.on_commit(idx)
        elif op == "insert":
            ii = step["index"]
            self.on_insert(self.indexes[ii], [self._insert_host(ii, d) for d in step["documents"]])
        elif op == "update_merge":  # UpdateDocumentRequest strategy merge (src/tests/update_docs.rs:55-75): old fields kept, given ones replaced
            ii = step["index"]
            merged = [{**self.docs[ii][d["id"]], **d} for d in step["documents"]]
            self.on_insert(self.indexes[ii], [self._insert_host(ii, d) for d in merged])
        elif op == "delete_index":
            self.deleted_indexes.add(step["index"])
        elif op == "delete":
            ii = step["index"]
            self.indexes[ii].delete_documents([self.ids[ii][s] for s in step["ids"] if s in self.ids[ii]])
        else:
            raise ValueError(op)

    def string_id(self, doc_id):
        for m in self.ids:
            for sid, n in m.items():
                if n == doc_id:
                    return sid
        raise KeyError(doc_id)

    def searched_indexes(self, p):
        """`indexes` of a request: absent or empty = every index of the collection; an ordinal the collection does not hold is
        ReadError::UnknownIndex (collection level, before any index is asked)."""
        want = p.get("indexes") or list(range(len(self.indexes)))
        for ii in want:
            if not (isinstance(ii, int) and 0 <= ii < len(self.indexes)):
                raise UnknownIndex(ii)
        # a DELETED index named by the request passes the validation and is skipped by the search loop
        # (src/tests/multi_index.rs:278-348: the reference accepts that outcome)
        return [ii for ii in want if ii not in self.deleted_indexes]

    # ---- the oracle's answer: (hits [(doc, score)], count) over all indexes
    def oracle_search(self, p):
        where = p.get("where")
        want = self.searched_indexes(p)
        check_filter_fields([self.indexes[ii] for ii in want], where)
        docs, scores = [], []
        for ii in want:
            idx = self.indexes[ii]
            if idx.document_count == 0:
                continue
            q = {k: v for k, v in p.items() if k not in ("where", "indexes")}
            if "properties" in q:
                q["properties"] = [n for n in q["properties"] if n in self.fields[ii]]
            allowed = FilterContext(idx).allowed_set(where)
            d, s = refcases.oracle_search(idx, q, self.fields[ii], allowed=allowed)
            docs.extend(int(x) for x in d)
            scores.extend(s)
        count = len(docs)
        lim, off = p.get("limit", 10), p.get("offset", 0)
        if not docs:
            return [], 0
        td, ts = orc.top_n(np.asarray(docs, dtype=np.uint64), np.asarray(scores, dtype=np.float32), lim + off)
        return [(int(a), float(b)) for a, b in zip(td[off:], ts[off:])], count


def check_expect(col, hits, count, exp):
    if "count" in exp:
        assert count == exp["count"], (count, exp)
    if "n_hits" in exp:
        assert len(hits) == exp["n_hits"], (len(hits), exp)
    if "ids" in exp:
        assert [col.string_id(h[0]) for h in hits] == exp["ids"]
    if "hit_ids_mod" in exp:
        m, r = exp["hit_ids_mod"]
        assert all(int(col.string_id(h[0])) % m == r for h in hits), hits
    if "min_score_gt" in exp:
        assert all(h[1] > exp["min_score_gt"] for h in hits), hits
    if "score_ratio" in exp:  # assert_approx_eq!(a.score, b.score * value, tol)
        r = exp["score_ratio"]
        by_id = {col.string_id(h[0]): h[1] for h in hits}
        assert abs(by_id[r["num"]] - by_id[r["den"]] * r["value"]) < r["tol"], (by_id, r)


def run_case(case, make_collection, search):
    """Every step of the case: mutations applied, searches answered by `search(col, params) -> (hits, count)` and checked against
    the reference's expectations AND the oracle's answer (ids in order, scores bit for bit, count)."""
    col = make_collection(case)
    for step in case["steps"]:
        if step["op"] != "search":
            col.apply(step)
            continue
        exp = step["expect"]
        if exp.get("error") in ("FilterFieldNotFound", "UnknownIndex"):
            kind = FilterFieldNotFound if exp["error"] == "FilterFieldNotFound" else UnknownIndex
            for fn in (lambda: col.oracle_search(step["params"]), lambda: search(col, step["params"])):
                try:
                    fn()
                except kind:
                    continue
                raise AssertionError("expected " + exp["error"])
            continue
        o_hits, o_count = col.oracle_search(step["params"])
        check_expect(col, o_hits, o_count, exp)
        hits, count = search(col, step["params"])
        check_expect(col, hits, count, exp)
        assert count == o_count and [h[0] for h in hits] == [h[0] for h in o_hits], (step["params"], hits, o_hits)
        assert np.array_equal(np.array([h[1] for h in hits], dtype=np.float32).view(np.uint32),
                              np.array([h[1] for h in o_hits], dtype=np.float32).view(np.uint32)), (step["params"], hits, o_hits)
    return col
