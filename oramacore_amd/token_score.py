"""Host-side mirror of the scoring dispatcher and of the minimal index state it reads.

Mirrors (names / argument meaning) `TokenScoreContext` — src/collection_manager/sides/read/index/token_score.rs:
`search_full_text` :186-303, `search_vector` :309-351, `search_hybrid` :357-387, `execute` :460-509 — and the
hot part of `search_on_indexes` (src/collection_manager/sides/read/search.rs:297-343, 481-500): filter →
token scores → OMC → count → top-(limit+offset) → skip/take.

Difference by design: the reference materialises the whole `HashMap<DocumentId, f32>` and selects later;
here scoring, OMC, count and top-k are one fused device pass, so `execute` returns `(hits, count)`.

What is NOT mirrored (third-party / out of scope): the real tokenizer + stemmer
(`oramacore_lib::nlp::TextParser`), the FST dictionary with Levenshtein expansion, the embedding model
(PyO3 service) — `SimpleTokenizer`, `StringFieldStorage`'s prefix lookup and the injected `embed` callable
are test-scale stand-ins that feed the same interfaces.
"""
from __future__ import annotations

import re
from dataclasses import dataclass, field
from typing import Callable, Optional

import numpy as np

from .context import Context
from .embedding_field import AllowBitmap, EmbeddingFieldStorage, VectorSearchParams
from .fulltext import (TermDictionary, B_DEFAULT, K1_DEFAULT, FacetField, PostingList, PostingsStore, ScoreMap,
                       threshold_tokens)


# ----------------------------------------------------------------------------- modes (src/types.rs:838-940)
@dataclass
class FulltextMode:
    term: str
    threshold: Optional[float] = None  # Threshold(f32), types.rs:860
    exact: bool = False
    tolerance: Optional[int] = None


@dataclass
class VectorMode:
    term: str
    similarity: float = 0.7  # Similarity default, types.rs:879-885


@dataclass
class HybridMode:
    term: str
    similarity: float = 0.7
    threshold: Optional[float] = None
    exact: bool = False
    tolerance: Optional[int] = None


class SimpleTokenizer:
    """Stand-in for nlp::TextParser::tokenize_and_stem: lowercase alphanumeric runs, no stemming
    (returns (token, None) pairs like the reference's (String, Option<String>))."""

    _re = re.compile(r"[0-9a-z]+")

    def tokenize_and_stem(self, text: str):
        return [(t, None) for t in self._re.findall(text.lower())]


class StringFieldStorage:
    """Host-side builder of one string field's postings (what `StringFieldStorage::insert`,
    index/string_field.rs:155-176, feeds the third-party store): term → (doc, tf), field_length per doc."""

    def __init__(self, tokenizer=None):
        self.tokenizer = tokenizer or SimpleTokenizer()
        self.postings: dict[str, dict[int, int]] = {}
        self.field_len: dict[int, int] = {}

    def insert(self, doc_id: int, text: str) -> None:
        toks = [t for t, _ in self.tokenizer.tokenize_and_stem(text)]
        self.field_len[doc_id] = min(len(toks), 0xFFFF)  # field_length: u16
        for t in toks:
            d = self.postings.setdefault(t, {})
            d[doc_id] = min(d.get(doc_id, 0) + 1, 0xFFFF)

    def delete(self, doc_id: int) -> None:
        self.field_len.pop(doc_id, None)
        for d in self.postings.values():
            d.pop(doc_id, None)

    def avg_field_length(self) -> float:
        return float(np.float32(np.mean(list(self.field_len.values())))) if self.field_len else 1.0


@dataclass
class Index:
    """The slice of the read-side `Index` the hot path reads (IndexSearchStore, index/mod.rs:88-105):
    document_count, string fields, embedding fields, OMC map."""

    ctx: Context
    string_fields: dict[int, StringFieldStorage] = field(default_factory=dict)
    embedding_fields: dict[int, EmbeddingFieldStorage] = field(default_factory=dict)
    omc: dict[int, float] = field(default_factory=dict)
    document_ids: set = field(default_factory=set)
    # field path -> (FieldId, type name) — path_to_field_id_map (token_score.rs:140-176): a request names its properties, every
    # index resolves them against its OWN fields and ignores the names it does not hold as string fields
    path_to_field_id_map: dict = field(default_factory=dict)
    # filter fields (index/{bool,number,string_filter}_field.rs) as the facet / group code reads them:
    # field name -> {doc id: value | [values]}
    bool_fields: dict = field(default_factory=dict)
    number_fields: dict = field(default_factory=dict)
    string_filter_fields: dict = field(default_factory=dict)
    date_fields: dict = field(default_factory=dict)  # timestamps (ms): a string value that parses as a date (filter.py::parse_date)
    # deletes since the last commit: every search carries NOT(deleted) until then (index/mod.rs:1346-1424, filter.rs:352-390)
    uncommitted_deleted_documents: set = field(default_factory=set)
    _post: Optional[PostingsStore] = None
    _lists: dict = field(default_factory=dict)      # (field_id, term) -> [list ids]: the committed list, then delta lists
    _terms: dict = field(default_factory=dict)      # field_id -> sorted term list
    _delta_terms: dict = field(default_factory=dict)  # field_id -> terms that exist only in delta lists (not in the dictionary)

    @property
    def document_count(self) -> int:
        return len(self.document_ids)

    def commit(self) -> None:
        """Export the committed postings to HBM (the GPU side of `compact`, INTEGRATION.md §3 ii)."""
        docs = np.array(sorted(self.document_ids), dtype=np.uint64)
        field_ids = sorted(self.string_fields)
        lists, self._lists, self._terms, self._delta_terms = [], {}, {}, {}
        self.uncommitted_deleted_documents = set()  # the rebuild holds live documents only
        if len(docs) == 0:  # an index without documents holds nothing resident (searches skip it: document_count == 0)
            if self._post is not None:
                self._post.close()
                self._post = None
            return
        for fi, fid in enumerate(field_ids):
            sf = self.string_fields[fid]
            self._terms[fid] = sorted(sf.postings)
            for term in self._terms[fid]:
                pl = sorted((d, tf) for d, tf in sf.postings[term].items() if d in self.document_ids)
                if not pl:
                    continue
                d = np.array([x[0] for x in pl], dtype=np.uint64)
                self._lists[(fid, term)] = [len(lists)]
                lists.append(PostingList(field=fi, docs=d, tf=np.array([x[1] for x in pl]),
                                         field_len=np.array([sf.field_len[int(x)] for x in d])))
        if self._post is None:
            self._post = PostingsStore(self.ctx)
        self._post.build(docs, [self.string_fields[f].avg_field_length() for f in field_ids], lists)
        if self.omc:
            self._post.set_omc(self.omc)
        self._field_order = field_ids
        # resident term dictionaries: the non-exact dictionary step (prefix / Levenshtein) runs on the device
        for d in getattr(self, "_dicts", {}).values():
            d.close()
        self._dicts = {fid: TermDictionary(self.ctx, sorted(self._terms[fid], key=lambda t: t.encode("utf-8")))
                       for fid in field_ids if self._terms.get(fid)}
        self._commit_filter_fields()

    # ---- live updates between commits
    def append_documents(self, doc_ids) -> None:
        """`StringFieldStorage::insert` between commits (string_field.rs:155-170; update_data(&self), index/mod.rs:1436): the
        documents `doc_ids` — already inserted into this mirror's host-side fields and `document_ids`, ids greater than every
        stored one — reach the device as DELTA posting lists (orama_post_append); a term is then its committed list plus
        its delta lists, passed with the same token.  The field averages move with the insert."""
        new = sorted(int(d) for d in doc_ids)
        if not new:
            return
        assert self._post is not None, "append_documents needs a committed index (commit() first)"
        field_ids = self._field_order
        assert sorted(self.string_fields) == field_ids, "a new string field needs a commit"
        lists, keys = [], []
        newset = set(new)
        for fi, fid in enumerate(field_ids):
            sf = self.string_fields[fid]
            for term in sorted(sf.postings):
                pl = sorted((d, tf) for d, tf in sf.postings[term].items() if d in newset)
                if not pl:
                    continue
                d = np.array([x[0] for x in pl], dtype=np.uint64)
                keys.append((fid, term))
                lists.append(PostingList(field=fi, docs=d, tf=np.array([x[1] for x in pl]),
                                         field_len=np.array([sf.field_len[int(x)] for x in d])))
        first = self._post.append(np.array(new, dtype=np.uint64), [self.string_fields[f].avg_field_length() for f in field_ids], lists)
        for i, key in enumerate(keys):
            if key not in self._lists:
                self._delta_terms.setdefault(key[0], set()).add(key[1])
            self._lists.setdefault(key, []).append(first + i)
        if self.omc:
            self._post.set_omc({d: m for d, m in self.omc.items() if d in self.document_ids or d in self.uncommitted_deleted_documents})

    def delete_documents(self, doc_ids) -> None:
        """IndexWriteOperation::DeleteDocuments (index/mod.rs:1346-1424): out of every field, the OMC map and the document
        count; remembered in `uncommitted_deleted_documents`, which turns every later search's filter into ... AND NOT(deleted)
        (filter.rs:352-390) — on the device the postings stay in their lists until the next commit and the bitmap hides them."""
        for d in (int(x) for x in doc_ids):
            if d not in self.document_ids:
                continue
            self.document_ids.discard(d)
            self.uncommitted_deleted_documents.add(d)
            for sf in self.string_fields.values():
                sf.delete(d)
            for store in (self.bool_fields, self.number_fields, self.string_filter_fields, self.date_fields):
                for vals in store.values():
                    vals.pop(d, None)
            self.omc.pop(d, None)
        if self._post is not None:  # StringStorage::info().avg_field_length follows the live documents
            self._post.set_avg_len([self.string_fields[f].avg_field_length() for f in self._field_order])

    def calculate_filter(self, where: dict | None):
        """Index::calculate_filter -> FilterContext::execute_filter (filter.rs:344-392): the AllowBitmap a search of this index
        carries, or None (no `where`, no uncommitted deletes)."""
        from .filter import FilterContext

        hi = max(list(self.document_ids) + list(self.uncommitted_deleted_documents) + [0]) + 1
        return FilterContext(self).execute_filter(where, n_bits=hi)

    # ---- resident images of the filter fields (orama_facet_field), rebuilt at commit like the postings
    def _commit_filter_fields(self) -> None:
        for f in getattr(self, "_facet_fields", {}).values():
            f[0].close()
        self._facet_fields = {}
        live = self.document_ids
        for name, vals in self.bool_fields.items():
            keys = [True, False]  # BoolFacetDefinition {true, false}, bool_field.rs:182-208
            buckets = [sorted(d for d, v in vals.items() if v is k and d in live) for k in keys]
            self._facet_fields[name] = (FacetField.buckets(self._post, buckets), "bool", keys)
        for name, vals in self.string_filter_fields.items():
            by_key: dict = {}
            for d, v in vals.items():
                if d not in live:
                    continue
                for key in (v if isinstance(v, (list, tuple)) else [v]):
                    by_key.setdefault(key, set()).add(d)
            keys = sorted(by_key)  # storage.keys(), string_filter_field.rs:180
            self._facet_fields[name] = (FacetField.buckets(self._post, [sorted(by_key[k]) for k in keys]), "string", keys)
        for name, vals in self.number_fields.items():
            docs, nums = [], []
            for d, v in vals.items():
                if d not in live:
                    continue
                for x in (v if isinstance(v, (list, tuple)) else [v]):
                    docs.append(d)
                    nums.append(float(x))
            self._facet_fields[name] = (FacetField.numbers(self._post, docs, nums), "number", None)

    def group_variants(self, name: str) -> dict:
        """calculate_group_for_field (group.rs:181-270): value -> set of doc ids."""
        live = self.document_ids
        out: dict = {}
        for store in (self.bool_fields, self.number_fields, self.string_filter_fields):
            if name in store:
                for d, v in store[name].items():
                    if d not in live:
                        continue
                    for x in (v if isinstance(v, (list, tuple)) else [v]):
                        out.setdefault(x, set()).add(d)
                return out
        return None

    def lookup(self, field_id: int, token: str, exact: bool, tolerance: int | None = None) -> list[int]:
        """Dictionary step of collect_contributions (host side, third-party in the reference): the exact term, or
        (non-exact) every term with the token as prefix — src/tests/fulltext_search.rs:603-753 ("christoph" matches
        "christopher") — plus, with `tolerance`, every term within that Levenshtein distance of the token —
        src/tests/fulltext_search.rs:956-1018 ("Mxin" finds "Main Street" with tolerance 1).  List order: dictionary
        (lexicographic) order of the terms."""
        return [l for l, _ in self.lookup_terms(field_id, token, exact, tolerance)]

    def lookup_terms(self, field_id: int, token: str, exact: bool, tolerance: int | None = None) -> list[tuple]:
        """`lookup` with the match kind: [(list id, term == token)] — the caller folds the exact-match boost into the
        reference's `boost` of the lists whose term IS the token (SURVEY §8c assumption 3)."""
        if exact:
            return [(l, True) for l in self._lists.get((field_id, token), [])]
        d = self._dicts.get(field_id) if hasattr(self, "_dicts") else None
        terms = [] if d is None else [d.terms[ti] for ti in d.expand(token, exact=False, tolerance=int(tolerance or 0))]
        # terms that exist only in delta lists are not in the resident dictionary (it is rebuilt at commit): the same rule on the host
        tol = int(tolerance or 0)
        extra = [t for t in self._delta_terms.get(field_id, ()) if t.startswith(token) or (tol and _levenshtein_le(t, token, tol))]
        out = []
        for term in sorted(set(terms) | set(extra), key=lambda t: t.encode("utf-8")):  # ascending = dictionary order
            for l in self._lists.get((field_id, term), []):
                out.append((l, term == token))
        return out


def _levenshtein_le(a: str, b: str, k: int) -> bool:
    """Levenshtein(a, b) <= k over bytes (the dictionary's unit)."""
    a, b = a.encode("utf-8"), b.encode("utf-8")
    if abs(len(a) - len(b)) > k:
        return False
    prev = list(range(len(b) + 1))
    for i, ca in enumerate(a, 1):
        cur = [i] + [0] * len(b)
        for j, cb in enumerate(b, 1):
            cur[j] = min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (ca != cb))
        if min(cur) > k:
            return False
        prev = cur
    return prev[-1] <= k


@dataclass
class TokenScoreParams:
    mode: object
    properties: Optional[list] = None            # string field ids or NAMES (resolved per index); None = all (Properties::None | Star)
    boost: dict = field(default_factory=dict)    # field_id -> boost
    limit: int = 10                              # Limit default 10, types.rs:748-754
    offset: int = 0
    # `limit_hint` (token_score.rs:488,497 <- search.rs:334 `limit_hint: score_params.limit`): the k of the vector
    # leg is the REQUEST's limit, never limit + offset; None = `limit`
    limit_hint: Optional[int] = None
    filtered_doc_ids: Optional[AllowBitmap] = None
    # the request's `where` (types.rs WhereFilter as JSON): search_on_indexes evaluates it PER INDEX (search.rs:304-338 ->
    # Index::calculate_filter) into that index's filtered_doc_ids, together with its uncommitted deletes
    where_filter: Optional[dict] = None


# The exact-match factor of the third-party string store (oramacore_fields 0.2.0, folded into ntf: token_score.rs:182-185,
# 226-228, 268-269).  Its value is NOT in the checkout.  What the reference does pin: boost_integration.rs:449-491 asserts
# hits[0].score > hits[1].score for "serve" against "server" — with a factor of 1.0 the two scores are bit-equal, so the
# factor is > 1.  1.5 is a PLACEHOLDER that satisfies every relation the reference's tests hold; the Rust shim passes the real
# one (AcceleratorConfig.exact_match_boost, "read it from oramacore_fields::string::SearchParams'
# implementation when the crate is available").  include/orama/host.hpp carries the same constant.
DEFAULT_EXACT_MATCH_BOOST = 1.5


class TokenScoreContext:
    def __init__(self, index: Index, embed: Callable[[str, object], np.ndarray] | None = None, tokenizer=None,
                 exact_match_boost: float = DEFAULT_EXACT_MATCH_BOOST):
        self.index = index
        self.embed = embed
        self.text_parser = tokenizer or SimpleTokenizer()
        # The third-party store folds an exact-match boost into ntf (comments token_score.rs:182-185, 226-228); its
        # value is not visible in this checkout (SURVEY §8c assumption 3), so it is a parameter: the factor applied
        # to postings of the dictionary term that equals the query token (prefix / fuzzy expansions keep 1.0).  It
        # reaches the device through orama_term_ref.boost (= field boost x this factor).  Default: the placeholder above
        # (> 1, as the reference's test_boost_match_exact requires); 1.0 = the bit-equal variant.
        self.exact_match_boost = float(exact_match_boost)

    # token_score.rs:196-209
    def _tokens(self, term: str, exact: bool) -> list[str]:
        toks = self.text_parser.tokenize_and_stem(term)
        out = [t for t, _ in toks] if exact else [x for t, s in toks for x in ((t,) if s is None else (t, s))]
        return out or [""]

    def calculate_string_properties(self, properties) -> list[int]:
        """token_score.rs:154-177: Properties::None | Star = every string field of the index; Specified = the given fields this
        index holds as STRING fields (names through path_to_field_id_map; unknown names and other types are skipped — an index that
        holds none of them searches nothing, src/tests/fulltext_search.rs:1021-1105).  Canonical order: ascending FieldId."""
        if properties is None:
            return sorted(self.index.string_fields)
        out = set()
        for f in properties:
            if isinstance(f, str):
                hit = self.index.path_to_field_id_map.get(f)
                if hit is None or hit[1] != "string":
                    continue
                f = hit[0]
            if f in self.index.string_fields:
                out.add(f)
        return sorted(out)

    def _refs(self, tokens, properties, boost, exact, tolerance=None):
        fields = self.calculate_string_properties(properties)
        refs = []
        for ti, tok in enumerate(tokens):
            for fid in fields:
                for l, is_exact_term in self.index.lookup_terms(fid, tok, exact, tolerance):
                    bo = np.float32(boost.get(fid, 1.0))
                    if is_exact_term and self.exact_match_boost != 1.0:
                        bo = np.float32(bo * np.float32(self.exact_match_boost))
                    refs.append((ti, l, float(bo)))
        return refs

    # token_score.rs:186-303 (+ OMC, count, top-(limit+offset) fused)
    def search_full_text(self, mode: FulltextMode, params: TokenScoreParams, vector: dict | None = None):
        tokens = self._tokens(mode.term, mode.exact)
        thr = None if mode.threshold is None else threshold_tokens(len(tokens), mode.threshold)
        refs = self._refs(tokens, params.properties, params.boost, mode.exact, mode.tolerance)
        return self.index._post.search(refs, len(tokens), float(self.index.document_count),
                                       params.limit + params.offset, thr, allow=params.filtered_doc_ids,
                                       apply_omc=bool(self.index.omc), b=B_DEFAULT, k=K1_DEFAULT, vector=vector)

    # token_score.rs:309-351 — returns the map after the a2 epilogue
    def search_vector(self, mode, params: TokenScoreParams) -> dict:
        output: dict = {}
        for fid in sorted(self.index.embedding_fields):
            ef = self.index.embedding_fields[fid]
            target = self.embed(mode.term, ef.model())
            limit_hint = params.limit if params.limit_hint is None else params.limit_hint
            ef.search(VectorSearchParams(target=target, similarity=mode.similarity, limit=limit_hint,
                                         filtered_doc_ids=params.filtered_doc_ids), output)
        return output

    # token_score.rs:460-509 + search.rs:342-343, 481-500
    def execute(self, params: TokenScoreParams):
        """Returns (hits [(doc_id, score)] after skip(offset).take(limit), count)."""
        from . import fulltext as ft

        if params.filtered_doc_ids is None and (params.where_filter is not None or self.index.uncommitted_deleted_documents):
            # Index::calculate_filter (filter.rs:344-392): the request's `where` and NOT(uncommitted deletes), as this index sees them
            from dataclasses import replace
            params = replace(params, filtered_doc_ids=self.index.calculate_filter(params.where_filter))
        m = params.mode
        top = params.limit + params.offset
        if isinstance(m, FulltextMode):
            ids, sc, count = self.search_full_text(m, params)
        elif isinstance(m, VectorMode):
            vec = self.search_vector(m, params)
            if self.index.omc:
                vec = {d: np.float32(s * np.float32(self.index.omc[d])) if d in self.index.omc else s
                       for d, s in vec.items()}
            ids, sc = ft.top_n(self.index.ctx, vec, top) if vec else (np.zeros(0, np.uint64), np.zeros(0, np.float32))
            count = len(vec)
        elif isinstance(m, HybridMode):
            vec = self.search_vector(m, params)
            ids, sc, count = self.search_full_text(
                FulltextMode(m.term, m.threshold, m.exact, m.tolerance), params, vector=vec)
        else:
            raise TypeError(f"unknown score mode {type(m)}")
        hits = list(zip(ids.tolist(), sc.tolist()))[params.offset: params.offset + params.limit]
        return hits, count


def facets_and_groups(tsc: "TokenScoreContext", params: TokenScoreParams, facets: dict | None = None,
                      group_by: tuple | None = None, has_where_filter: bool = False, not_deleted=None):
    """The facet / group tail of `search_on_indexes` for ONE index (src/collection_manager/sides/read/search.rs:
    345-420 + sort.rs:129-230 without sort_by), over the score map kept resident in HBM.

    facets   : {field name: "bool" | "string" | [(from, to), ...]}  (FacetDefinition, types.rs:770-800)
    group_by : (properties [field names], max_results)
    has_where_filter : the request carried a `where` filter — the reference then RE-SCORES without it (only the
        NOT-deleted predicate, `not_deleted`) and counts facets on that map, so that the facet numbers do not move when
        a user clicks a category (search.rs:347-396).  Groups always use the filtered map (sort.rs:129-136).
    Returns (hits, count, facet_results {name: {"count", "values"}}, group_results {tuple(values): [(doc, score)]})."""
    m = params.mode
    if not isinstance(m, FulltextMode):
        raise TypeError("facets_and_groups mirrors the full-text path (the vector map is <= limit entries on the host)")
    idx = tsc.index
    if idx.document_count == 0:
        # an index without documents scores nothing, holds no filter field and forms no group (src/tests/groupby.rs:950-982,
        # multi_index.rs:88-167): nothing to launch
        return [], 0, {}, {}
    tokens = tsc._tokens(m.term, m.exact)
    thr = None if m.threshold is None else threshold_tokens(len(tokens), m.threshold)
    refs = tsc._refs(tokens, params.properties, params.boost, m.exact, m.tolerance)
    top = params.limit + params.offset

    def scored(allow) -> ScoreMap:
        return idx._post.search_scores(refs, len(tokens), float(idx.document_count), top, thr, allow=allow,
                                       apply_omc=bool(idx.omc))

    sm = scored(params.filtered_doc_ids)
    ids, sc, count = sm.hits
    hits = list(zip(ids.tolist(), sc.tolist()))[params.offset: params.offset + params.limit]
    facet_results, group_results = {}, {}
    try:
        if facets:
            fm = scored(not_deleted) if has_where_filter else sm
            try:
                for name, definition in facets.items():
                    if name not in idx._facet_fields:
                        continue  # warn!("Unknown field name"), facet.rs:159-163
                    fld, kind, keys = idx._facet_fields[name]
                    if kind == "number":
                        counts = fm.facet_count_ranges(fld, definition)
                        values = {f"{a}-{b}": int(c) for (a, b), c in zip(definition, counts)}  # number_field.rs:382
                    else:
                        counts = fm.facet_count(fld)
                        labels = [("true" if k else "false") for k in keys] if kind == "bool" else keys
                        values = {str(l): int(c) for l, c in zip(labels, counts)}
                    facet_results[name] = {"count": len(values), "values": values}  # facet.rs:196-206
            finally:
                if fm is not sm:
                    fm.close()
        if group_by:
            props, max_results = group_by
            variants = [idx.group_variants(p) for p in props]
            if all(v is not None for v in variants):  # group.rs:113-121: every property must exist in this index
                import itertools

                combos, buckets = [], []
                for combo in itertools.product(*[sorted(v, key=str) for v in variants]):
                    docs = set.intersection(*[variants[i][val] for i, val in enumerate(combo)])  # group.rs:143-160
                    combos.append(combo)
                    buckets.append(sorted(docs))
                if max_results == 0:  # (src/tests/groupby.rs:624-663: the groups exist and are empty — nothing to select)
                    for combo in combos:
                        group_results[combo] = []
                else:
                    fld = FacetField.buckets(idx._post, buckets)
                    try:
                        g_ids, g_sc, g_n = sm.group_top(fld, max_results)  # sort.rs:203-213
                    finally:
                        fld.close()
                    for combo, i_, s_, n_ in zip(combos, g_ids, g_sc, g_n):
                        group_results[combo] = list(zip(i_[: int(n_)].tolist(), s_[: int(n_)].tolist()))
    finally:
        sm.close()
    return hits, count, facet_results, group_results


class FacetFieldNotFound(KeyError):
    """ReadError::FacetFieldNotFound (read/mod.rs:148)."""


def check_facet_results(requested, facets_results: dict) -> None:
    """search.rs:452-463: after the facets of every searched index were added up, a requested facet that NO index produced (each
    index skips the names it does not hold, facet.rs:159-163) fails the search with the missing names."""
    missing = [name for name in requested if name not in facets_results]
    if missing:
        raise FacetFieldNotFound(missing)


def search_on_indexes(contexts: list, params: TokenScoreParams):
    """`Search::search_on_indexes` + the tail of `Search::execute` (src/collection_manager/sides/read/search.rs:
    297-343, 481-500) over several indexes of one collection: every index scores on its own (BM25 statistics, the
    hybrid min-max and the OMC multipliers are per index — token_score.rs:221, 492-500; index/mod.rs:1720-1739),
    the per-index score maps are `extend`ed into one (DocumentIds are collection-unique, so nothing merges),
    `count` is the size of that map and the hits are its top-(limit + offset) after skip(offset).take(limit).

    The per-index maps never leave the GPU: an index hands back its own top-(limit + offset) and its map size; the
    global top-(limit + offset) of a disjoint union is contained in the union of the per-index tops, so the final
    cut over those (K4 again, `orama_top_n`) is exact."""
    from . import fulltext as ft

    from .filter import check_filter_fields

    top = params.limit + params.offset
    check_filter_fields([t.index for t in contexts], params.where_filter)  # ReadError::FilterFieldNotFound, search.rs:435-449
    docs, scores, count = [], [], 0
    for tsc in contexts:
        # every index returns its own top-(limit + offset); the vector leg keeps the request's limit (limit_hint,
        # search.rs:334 / token_score.rs:339-344) — with offset > 0 the two differ.  The filter is the index's own: its fields,
        # its uncommitted deletes (a caller-made bitmap, if any, is AND-ed by passing it as filtered_doc_ids of a one-index call)
        allow = params.filtered_doc_ids
        if params.where_filter is not None or tsc.index.uncommitted_deleted_documents:
            assert allow is None, "pass either a where filter or a ready bitmap"
            allow = tsc.index.calculate_filter(params.where_filter)
        per_index = TokenScoreParams(mode=params.mode, properties=params.properties, boost=params.boost, limit=top, offset=0,
                                     filtered_doc_ids=allow,
                                     limit_hint=params.limit if params.limit_hint is None else params.limit_hint)
        if tsc.index.document_count == 0 or tsc.index._post is None:
            continue  # an index without documents contributes nothing (src/tests/multi_index.rs:88-167)
        hits, c = tsc.execute(per_index)
        count += c
        docs.extend(h[0] for h in hits)
        scores.extend(h[1] for h in hits)
    if not docs:
        return [], count
    ids, sc = ft.top_n(contexts[0].index.ctx, (np.asarray(docs, dtype=np.uint64), np.asarray(scores, dtype=np.float32)),
                       top)
    hits = list(zip(ids.tolist(), sc.tolist()))[params.offset: params.offset + params.limit]
    return hits, count
