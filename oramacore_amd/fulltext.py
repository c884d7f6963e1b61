"""Host-side mirror of the reference's full-text / hybrid scoring interfaces over the C ABI.

Reference interfaces mirrored (names, argument meaning, error behaviour):
  * `BM25Scorer` — src/collection_manager/bm25.rs:135-323 (plain / with_threshold, add_precomputed_field,
    finalize_term, get_scores).  The reference scores on the CPU while contributions are added; here the
    calls only record the contributions and `get_scores()/top_n()` runs K3 (+K4) on the GPU through
    `orama_bm25_score`.
  * `normalize_and_combine` — token_score.rs:393-422  → `orama_hybrid_combine`
  * `top_n` — sort.rs:260-279                          → `orama_top_n`
  * `PostingsStore` — the HBM-resident replacement of what `StringFieldStorage`
    (index/string_field.rs) holds, searched with `search_full_text` / `search_hybrid`
    (token_score.rs:186-303, 357-387) semantics through `orama_post_search[_hybrid]`.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field

import numpy as np

from . import _native as N
from .context import Context, Owner
from .embedding_field import AllowBitmap

K1_DEFAULT = 1.2  # token_score.rs:283
B_DEFAULT = 0.75  # Bm25Params::default()


def _u64(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _params(total_documents: float, n_tokens: int, threshold, top_k: int, k: float = K1_DEFAULT) -> N.Bm25Params:
    p = N.Bm25Params()
    p.total_documents = float(total_documents)
    p.k = float(k)
    p.n_tokens = int(n_tokens)
    p.use_threshold = 0 if threshold is None else 1
    p.threshold = 0 if threshold is None else int(threshold)
    p.top_k = int(top_k)
    return p


def threshold_tokens(n_tokens: int, threshold: float) -> int:
    """token_score.rs:213-214 — floor(tokens.len() as f32 * threshold) as u32."""
    return int(np.floor(np.float32(n_tokens) * np.float32(threshold)))


def bm25_score(ctx: Context, entries, n_tokens: int, total_documents: float, top_k: int, threshold=None,
               omc: dict | None = None, k: float = K1_DEFAULT):
    """Seam (i). entries: list of (token, doc_ids, ntf). Returns (ids, scores, count)."""
    lib = N.load()
    keep = []
    arr = (N.NtfEntry * max(len(entries), 1))()
    for i, (tok, docs, ntf) in enumerate(entries):
        docs, ntf = _u64(docs), _f32(ntf)
        keep.append((docs, ntf))
        arr[i].token = int(tok)
        arr[i].doc = docs.ctypes.data_as(C.POINTER(C.c_uint64))
        arr[i].ntf = ntf.ctypes.data_as(C.POINTER(C.c_float))
        arr[i].len = docs.shape[0]
    params = _params(total_documents, n_tokens, threshold, top_k, k)
    out_ids = np.zeros(max(top_k, 1), dtype=np.uint64)
    out_sc = np.zeros(max(top_k, 1), dtype=np.float32)
    out_n = C.c_uint32()
    out_count = C.c_uint64()
    omc_doc = omc_mul = None
    n_omc = 0
    if omc:
        items = sorted(omc.items())
        omc_doc = _u64([d for d, _ in items])
        omc_mul = _f32([m for _, m in items])
        n_omc = len(items)
    N.check(lib.orama_bm25_score(ctx.handle, arr, len(entries), C.byref(params),
                                 omc_doc.ctypes.data if n_omc else None, omc_mul.ctypes.data if n_omc else None,
                                 n_omc, out_ids.ctypes.data, out_sc.ctypes.data, C.byref(out_n), C.byref(out_count)))
    return out_ids[: out_n.value], out_sc[: out_n.value], out_count.value


def bm25_score_map(ctx: Context, entries, n_tokens: int, total_documents: float, threshold=None, omc: dict | None = None,
                   k: float = K1_DEFAULT) -> dict:
    """Seam (i), the WHOLE score map (BM25Scorer::get_scores, bm25.rs:416-428) — any size."""
    lib = N.load()
    keep = []
    arr = (N.NtfEntry * max(len(entries), 1))()
    for i, (tok, docs, ntf) in enumerate(entries):
        docs, ntf = _u64(docs), _f32(ntf)
        keep.append((docs, ntf))
        arr[i].token = int(tok)
        arr[i].doc = docs.ctypes.data_as(C.POINTER(C.c_uint64))
        arr[i].ntf = ntf.ctypes.data_as(C.POINTER(C.c_float))
        arr[i].len = docs.shape[0]
    params = _params(total_documents, n_tokens, threshold, 0, k)
    omc_doc = omc_mul = None
    n_omc = 0
    if omc:
        items = sorted(omc.items())
        omc_doc, omc_mul, n_omc = _u64([d for d, _ in items]), _f32([m for _, m in items]), len(items)
    cap = sum(len(e[1]) for e in entries)
    out_ids = np.zeros(max(cap, 1), dtype=np.uint64)
    out_sc = np.zeros(max(cap, 1), dtype=np.float32)
    out_n = C.c_uint64()
    N.check(lib.orama_bm25_score_map(ctx.handle, arr, len(entries), C.byref(params),
                                     omc_doc.ctypes.data if n_omc else None, omc_mul.ctypes.data if n_omc else None, n_omc,
                                     cap, out_ids.ctypes.data, out_sc.ctypes.data, C.byref(out_n)))
    return {int(d): np.float32(v) for d, v in zip(out_ids[: out_n.value], out_sc[: out_n.value])}


def hybrid_combine(ctx: Context, vector: dict, fulltext: dict, top_k: int):
    """normalize_and_combine (token_score.rs:393-422) + count + top_n. Returns (ids, scores, count)."""
    lib = N.load()
    v_doc, v_sc = _u64(list(vector.keys())), _f32(list(vector.values()))
    f_doc, f_sc = _u64(list(fulltext.keys())), _f32(list(fulltext.values()))
    out_ids = np.zeros(max(top_k, 1), dtype=np.uint64)
    out_sc = np.zeros(max(top_k, 1), dtype=np.float32)
    out_n = C.c_uint32()
    out_count = C.c_uint64()
    N.check(lib.orama_hybrid_combine(ctx.handle, v_doc.ctypes.data, v_sc.ctypes.data, v_doc.shape[0],
                                     f_doc.ctypes.data, f_sc.ctypes.data, f_doc.shape[0], top_k,
                                     out_ids.ctypes.data, out_sc.ctypes.data, C.byref(out_n), C.byref(out_count)))
    return out_ids[: out_n.value], out_sc[: out_n.value], out_count.value


def hybrid_rrf(ctx: Context, vector: dict, fulltext: dict, top_k: int, rrf_k: float = 60.0, depth: int = 1000):
    """Reciprocal-rank fusion (extra; the parity path is hybrid_combine). Returns (ids, scores, count)."""
    lib = N.load()
    v_doc, v_sc = _u64(list(vector.keys())), _f32(list(vector.values()))
    f_doc, f_sc = _u64(list(fulltext.keys())), _f32(list(fulltext.values()))
    out_ids = np.zeros(max(top_k, 1), dtype=np.uint64)
    out_sc = np.zeros(max(top_k, 1), dtype=np.float32)
    out_n = C.c_uint32()
    out_count = C.c_uint64()
    N.check(lib.orama_hybrid_rrf(ctx.handle, v_doc.ctypes.data, v_sc.ctypes.data, v_doc.shape[0], f_doc.ctypes.data,
                                 f_sc.ctypes.data, f_doc.shape[0], float(rrf_k), int(depth), top_k,
                                 out_ids.ctypes.data, out_sc.ctypes.data, C.byref(out_n), C.byref(out_count)))
    return out_ids[: out_n.value], out_sc[: out_n.value], out_count.value


def top_n(ctx: Context, token_scores: dict | tuple, n: int):
    """sort.rs:260-279 over a {DocumentId: score} map (or a (docs, scores) pair)."""
    lib = N.load()
    if isinstance(token_scores, dict):
        doc, sc = _u64(list(token_scores.keys())), _f32(list(token_scores.values()))
    else:
        doc, sc = _u64(token_scores[0]), _f32(token_scores[1])
    out_ids = np.zeros(max(n, 1), dtype=np.uint64)
    out_sc = np.zeros(max(n, 1), dtype=np.float32)
    out_n = C.c_uint32()
    N.check(lib.orama_top_n(ctx.handle, doc.ctypes.data, sc.ctypes.data, doc.shape[0], n, out_ids.ctypes.data,
                            out_sc.ctypes.data, C.byref(out_n)))
    return out_ids[: out_n.value], out_sc[: out_n.value]


class BM25Scorer:
    """Mirror of bm25.rs `BM25Scorer<K>` restricted to the calls search_full_text makes
    (token_score.rs:211-302): plain()/with_threshold(), reset_term(), add_precomputed_field(),
    finalize_term[_plain](), next_term(), get_scores()."""

    def __init__(self, ctx: Context, threshold: int | None):
        self.ctx = ctx
        self.threshold = threshold
        self._entries = []      # finalised (token, docs, ntf*weight)
        self._cur_docs, self._cur_ntf = [], []
        self._term_index = 0
        self._finalized = []    # (df, total_documents, k) per term, as passed by the caller

    @classmethod
    def plain(cls, ctx: Context) -> "BM25Scorer":
        return cls(ctx, None)

    @classmethod
    def with_threshold(cls, ctx: Context, threshold: int) -> "BM25Scorer":
        return cls(ctx, int(threshold))

    def reset_term(self) -> None:
        self._cur_docs, self._cur_ntf = [], []

    def add_precomputed_field(self, key: int, normalized_tf, weight=1.0) -> None:
        self._cur_docs.append(int(key))
        self._cur_ntf.append(np.float32(np.float32(weight) * np.float32(normalized_tf)))

    def current_term_document_count(self) -> int:
        return len(set(self._cur_docs))

    def finalize_term(self, corpus_term_frequency: int, total_documents: float, k: float, phrase_boost: float = 1.0,
                      token_indexes: int = 0) -> None:
        if phrase_boost != 1.0:
            raise N.OramaError(N.ORAMA_ERR_UNSUPPORTED, "phrase_boost != 1.0 (the reference always passes 1.0)")
        # One entry per field push: a doc may appear several times inside a term (one per field); split the
        # stream into runs with unique docs so that each run is a valid posting-list entry, in push order.
        runs, seen = [], set()
        cur_d, cur_v = [], []
        for d, v in zip(self._cur_docs, self._cur_ntf):
            if d in seen:
                runs.append((cur_d, cur_v))
                cur_d, cur_v, seen = [], [], set()
            seen.add(d)
            cur_d.append(d)
            cur_v.append(v)
        runs.append((cur_d, cur_v))
        for d, v in runs:
            self._entries.append((self._term_index, d, v))
        self._finalized.append((corpus_term_frequency, float(total_documents), float(k)))

    finalize_term_plain = finalize_term

    def next_term(self) -> None:
        self._term_index += 1
        self.reset_term()

    def _n_tokens(self) -> int:
        return max(self._term_index, len(self._finalized), 1)

    def top_n(self, n: int, omc: dict | None = None):
        """get_scores() + apply_omc_multipliers + count + top_n in one GPU pass."""
        total_documents = self._finalized[0][1] if self._finalized else 1.0
        k = self._finalized[0][2] if self._finalized else K1_DEFAULT
        return bm25_score(self.ctx, self._entries, self._n_tokens(), total_documents, n, self.threshold, omc, k)

    def get_scores(self) -> dict:
        """The whole HashMap<K, f32> (bm25.rs:416-428), any size (orama_bm25_score_map)."""
        total_documents = self._finalized[0][1] if self._finalized else 1.0
        k = self._finalized[0][2] if self._finalized else K1_DEFAULT
        return bm25_score_map(self.ctx, self._entries, self._n_tokens(), total_documents, self.threshold, None, k)


@dataclass
class PostingList:
    field: int
    docs: np.ndarray   # DocumentIds ascending
    tf: np.ndarray
    field_len: np.ndarray


class ScoreMap:
    """The HashMap<DocumentId, f32> of one search, resident in HBM (orama_scores): what the reference hands to
    FacetContext / GroupContext (search.rs:355-400).  Close it before the next build / append of its store."""

    def __init__(self, lib, handle, hits, store=None):
        self._lib = lib
        self._h = handle
        self.hits = hits  # (ids, scores, count) of the search that produced the map
        self._store = store  # keeps the store alive: the handle holds a read lock on it
        if store is not None:
            store._adopt(self)

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._lib.orama_scores_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def __len__(self) -> int:
        n = C.c_uint64()
        N.check(self._lib.orama_scores_count(self._h, C.byref(n)))
        return n.value

    def to_dict(self) -> dict:
        n = len(self)
        ids, sc, out_n = np.zeros(max(n, 1), dtype=np.uint64), np.zeros(max(n, 1), dtype=np.float32), C.c_uint64()
        N.check(self._lib.orama_scores_export(self._h, n, ids.ctypes.data, sc.ctypes.data, C.byref(out_n)))
        return {int(d): np.float32(v) for d, v in zip(ids[: out_n.value], sc[: out_n.value])}

    def lookup(self, doc_ids):
        d = _u64(doc_ids)
        sc = np.zeros(max(len(d), 1), dtype=np.float32)
        pr = np.zeros(max(len(d), 1), dtype=np.uint8)
        N.check(self._lib.orama_scores_lookup(self._h, d.ctypes.data, len(d), sc.ctypes.data, pr.ctypes.data))
        return sc[: len(d)], pr[: len(d)].astype(bool)

    def facet_count(self, field: "FacetField") -> np.ndarray:
        out = np.zeros(max(field.n_buckets, 1), dtype=np.uint64)
        N.check(self._lib.orama_facet_count(self._h, field._h, out.ctypes.data))
        return out[: field.n_buckets]

    def facet_count_ranges(self, field: "FacetField", ranges) -> np.ndarray:
        fr = np.ascontiguousarray([r[0] for r in ranges], dtype=np.float64)
        to = np.ascontiguousarray([r[1] for r in ranges], dtype=np.float64)
        out = np.zeros(max(len(ranges), 1), dtype=np.uint64)
        N.check(self._lib.orama_facet_count_ranges(self._h, field._h, fr.ctypes.data, to.ctypes.data, len(ranges),
                                                   out.ctypes.data))
        return out[: len(ranges)]

    def group_top(self, field: "FacetField", max_results: int):
        g = field.n_buckets
        ids = np.zeros((max(g, 1), max_results), dtype=np.uint64)
        sc = np.zeros((max(g, 1), max_results), dtype=np.float32)
        n = np.zeros(max(g, 1), dtype=np.uint32)
        N.check(self._lib.orama_group_top(self._h, field._h, int(max_results), ids.ctypes.data, sc.ctypes.data, n.ctypes.data))
        return ids[:g], sc[:g], n[:g]


class FacetField:
    """Resident image of one filter field of an index (orama_facet_field): buckets (bool / string filter / group
    combinations) or numbers.  index/{bool,string_filter,number}_field.rs."""

    def __init__(self, lib, handle, n_buckets, store=None):
        self._lib, self._h, self.n_buckets = lib, handle, n_buckets
        self._store = store  # a field must not outlive the store it was resolved against
        if store is not None:
            store._adopt(self)

    @classmethod
    def buckets(cls, store: "PostingsStore", buckets: list) -> "FacetField":
        """buckets: list of doc-id arrays, one per field value."""
        off = np.zeros(len(buckets) + 1, dtype=np.uint64)
        for i, b in enumerate(buckets):
            off[i + 1] = off[i] + len(b)
        docs = _u64(np.concatenate([_u64(b) for b in buckets]) if buckets else [])
        h = C.c_void_p()
        N.check(store._lib.orama_facet_field_create_buckets(store._h, off.ctypes.data, docs.ctypes.data, len(buckets),
                                                            C.byref(h)))
        return cls(store._lib, h, len(buckets), store)

    @classmethod
    def numbers(cls, store: "PostingsStore", docs, values) -> "FacetField":
        d = _u64(docs)
        v = np.ascontiguousarray(values, dtype=np.float64)
        h = C.c_void_p()
        N.check(store._lib.orama_facet_field_create_numbers(store._h, d.ctypes.data, v.ctypes.data, len(d), C.byref(h)))
        return cls(store._lib, h, 0, store)

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._lib.orama_facet_field_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


class PreparedBatch:
    """Marshalled arguments of one orama_post_search_batch call (see PostingsStore.prepare_batch)."""

    def __init__(self, store, queries, total_documents, top_k, allow, apply_omc, max_parallel, b, k):
        self._store = store
        nq = self.nq = len(queries)
        self._descs = (N.PostQueryDesc * max(nq, 1))()
        self._keep = []
        for i, qd in enumerate(queries):
            refs, n_tokens, thr = qd[0], qd[1], qd[2]
            arr = store._refs(refs)
            self._keep.append(arr)
            self._descs[i].refs = C.cast(arr, C.POINTER(N.TermRef))
            self._descs[i].n_refs = len(refs)
            self._descs[i].params = _params(total_documents, n_tokens, thr, qd[3] if len(qd) > 3 else top_k, k)
        self._stride = max(top_k, 1)
        self.out_ids = np.zeros((max(nq, 1), self._stride), dtype=np.uint64)
        self.out_sc = np.zeros((max(nq, 1), self._stride), dtype=np.float32)
        self.out_n = np.zeros(max(nq, 1), dtype=np.uint32)
        self.out_count = np.zeros(max(nq, 1), dtype=np.uint64)
        self._allow = allow
        self._args = (b, 1 if apply_omc else 0, int(max_parallel))

    def run(self, statuses: bool = False):
        """One library call.  With `statuses` nothing is raised for a failing query: returns the status array."""
        lib, st_ = self._store._lib, self._store
        bm_ptr, bm_bits = self._allow.ffi_args() if self._allow is not None else (None, 0)
        b, omc, par = self._args
        if statuses:
            st = np.zeros(max(self.nq, 1), dtype=np.int32)
            lib.orama_post_search_batch_status(st_._h, self._descs, self.nq, b, bm_ptr, bm_bits, omc, par, self._stride,
                                               self.out_ids.ctypes.data, self.out_sc.ctypes.data, self.out_n.ctypes.data,
                                               self.out_count.ctypes.data, st.ctypes.data)
            return st[: self.nq]
        N.check(lib.orama_post_search_batch(st_._h, self._descs, self.nq, b, bm_ptr, bm_bits, omc, par, self._stride,
                                            self.out_ids.ctypes.data, self.out_sc.ctypes.data, self.out_n.ctypes.data,
                                            self.out_count.ctypes.data))
        return None

    def results(self):
        return [(self.out_ids[i, : self.out_n[i]].copy(), self.out_sc[i, : self.out_n[i]].copy(), int(self.out_count[i]))
                for i in range(self.nq)]


class PostingsStore(Owner):
    """HBM-resident postings of one index (seam ii)."""

    def __init__(self, ctx: Context):
        self._lib = N.load()
        self.ctx = ctx
        h = C.c_void_p()
        N.check(self._lib.orama_post_create(ctx.handle, C.byref(h)))
        self._h = h
        self.n_docs = 0
        ctx._adopt(self)

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._close_children()  # score maps, facet fields, staged queries, batchers hold locks / pointers into it
            self._lib.orama_post_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def build(self, docs, avg_field_len, lists: list[PostingList]) -> None:
        docs = _u64(docs)
        avg = _f32(avg_field_len)
        n_lists = len(lists)
        fol = np.ascontiguousarray([l.field for l in lists], dtype=np.uint32)
        off = np.zeros(n_lists + 1, dtype=np.uint64)
        for i, l in enumerate(lists):
            off[i + 1] = off[i] + np.uint64(len(l.docs))
        if n_lists:
            pdoc = _u64(np.concatenate([_u64(l.docs) for l in lists]))
            ptf = np.ascontiguousarray(np.concatenate([np.asarray(l.tf) for l in lists]), dtype=np.uint32)
            plen = np.ascontiguousarray(np.concatenate([np.asarray(l.field_len) for l in lists]), dtype=np.uint32)
        else:
            pdoc, ptf, plen = _u64([]), np.zeros(0, np.uint32), np.zeros(0, np.uint32)
        N.check(self._lib.orama_post_build(self._h, docs.ctypes.data, docs.shape[0], avg.shape[0], avg.ctypes.data,
                                           n_lists, fol.ctypes.data, off.ctypes.data, pdoc.ctypes.data,
                                           ptf.ctypes.data, plen.ctypes.data))
        self.n_docs = int(docs.shape[0])

    def append(self, docs, avg_field_len, lists: list[PostingList]) -> int:
        """Live update (orama_post_append): new documents + delta posting lists. Returns the id of the first new list."""
        first = self.info()["n_lists"]
        docs = _u64(docs)
        avg = _f32(avg_field_len)
        n_lists = len(lists)
        fol = np.ascontiguousarray([l.field for l in lists], dtype=np.uint32)
        off = np.zeros(n_lists + 1, dtype=np.uint64)
        for i, l in enumerate(lists):
            off[i + 1] = off[i] + np.uint64(len(l.docs))
        if n_lists:
            pdoc = _u64(np.concatenate([_u64(l.docs) for l in lists]))
            ptf = np.ascontiguousarray(np.concatenate([np.asarray(l.tf) for l in lists]), dtype=np.uint32)
            plen = np.ascontiguousarray(np.concatenate([np.asarray(l.field_len) for l in lists]), dtype=np.uint32)
        else:
            pdoc, ptf, plen = _u64([]), np.zeros(0, np.uint32), np.zeros(0, np.uint32)
        N.check(self._lib.orama_post_append(self._h, docs.ctypes.data, docs.shape[0], avg.ctypes.data, n_lists,
                                            fol.ctypes.data, off.ctypes.data, pdoc.ctypes.data, ptf.ctypes.data,
                                            plen.ctypes.data))
        self.n_docs += int(docs.shape[0])
        return first

    def fill_synthetic(self, n_docs: int, ranks, seed: int, first_doc_id: int = 0) -> int:
        """Bench utility: Zipf(1.07) posting lists generated in HBM. Returns the number of postings."""
        r = np.ascontiguousarray(ranks, dtype=np.uint32)
        total = C.c_uint64()
        N.check(self._lib.orama_post_fill_synthetic(self._h, int(n_docs), int(first_doc_id), r.shape[0],
                                                    r.ctypes.data, int(seed), C.byref(total)))
        self.n_docs = int(n_docs)
        return total.value

    def hybrid_search(self, vec_store, query, limit: int, similarity: float, refs, n_tokens: int,
                      total_documents: float, top_k: int, threshold=None, allow: AllowBitmap | None = None,
                      apply_omc: bool = True, rescale_e5: bool = False, b: float = B_DEFAULT, k: float = K1_DEFAULT):
        """search_hybrid (token_score.rs:357-387) in one call: vector leg and full-text leg overlap on two HIP
        streams; epilogue, combine, OMC, count and top-k inside the library. Returns (ids, scores, count)."""
        arr = self._refs(refs)
        params = _params(total_documents, n_tokens, threshold, top_k, k)
        qv = _f32(query)
        out_ids = np.zeros(max(top_k, 1), dtype=np.uint64)
        out_sc = np.zeros(max(top_k, 1), dtype=np.float32)
        out_n = C.c_uint32()
        out_count = C.c_uint64()
        bm_ptr, bm_bits = allow.ffi_args() if allow is not None else (None, 0)
        N.check(self._lib.orama_hybrid_search(vec_store.handle, self._h, qv.ctypes.data, int(limit), float(similarity),
                                              1 if rescale_e5 else 0, arr, len(refs), b, C.byref(params), bm_ptr,
                                              bm_bits, 1 if apply_omc else 0, out_ids.ctypes.data, out_sc.ctypes.data,
                                              C.byref(out_n), C.byref(out_count)))
        return out_ids[: out_n.value], out_sc[: out_n.value], out_count.value

    def prepare_hybrid(self, vec_store, query, limit: int, similarity: float, refs, n_tokens: int, total_documents: float,
                       top_k: int, threshold=None, allow: AllowBitmap | None = None, apply_omc: bool = True,
                       rescale_e5: bool = False, b: float = B_DEFAULT, k: float = K1_DEFAULT) -> "PreparedHybrid":
        """The arguments of `hybrid_search` marshalled once (what a native caller holds anyway): `.run()` is the bare
        orama_hybrid_search call and returns what `hybrid_search` returns."""
        return PreparedHybrid(self, vec_store, query, limit, similarity, refs, n_tokens, total_documents, top_k, threshold, allow,
                              apply_omc, rescale_e5, b, k)

    def info(self) -> dict:
        nd, nl, npst, avg = C.c_uint64(), C.c_uint32(), C.c_uint64(), C.c_float()
        N.check(self._lib.orama_post_info(self._h, C.byref(nd), C.byref(nl), C.byref(npst), C.byref(avg)))
        return {"n_docs": nd.value, "n_lists": nl.value, "n_postings": npst.value, "avg_field_length": avg.value}

    def get_list(self, lst: int):
        """(docs, tf, field_len) of one posting list (checker utility)."""
        n = C.c_uint64()
        N.check(self._lib.orama_post_get_list(self._h, int(lst), 0, None, None, None, C.byref(n)))
        d = np.zeros(max(n.value, 1), dtype=np.uint64)
        tf = np.zeros(max(n.value, 1), dtype=np.uint32)
        ln = np.zeros(max(n.value, 1), dtype=np.uint32)
        N.check(self._lib.orama_post_get_list(self._h, int(lst), n.value, d.ctypes.data, tf.ctypes.data,
                                              ln.ctypes.data, C.byref(n)))
        return d[: n.value], tf[: n.value], ln[: n.value]

    def set_omc(self, omc: dict) -> None:
        items = sorted(omc.items())
        d, m = _u64([x for x, _ in items]), _f32([y for _, y in items])
        N.check(self._lib.orama_post_set_omc(self._h, d.ctypes.data, m.ctypes.data, len(items)))

    def _refs(self, refs):
        arr = (N.TermRef * max(len(refs), 1))()
        for i, (tok, lst, boost) in enumerate(refs):
            arr[i].token, arr[i].list, arr[i].boost = int(tok), int(lst), float(boost)
        return arr

    def search(self, refs, n_tokens: int, total_documents: float, top_k: int, threshold=None,
               allow: AllowBitmap | None = None, apply_omc: bool = True, b: float = B_DEFAULT,
               k: float = K1_DEFAULT, vector: dict | None = None):
        """refs: list of (token, list, boost). With `vector` (the map after the a2 epilogue) runs the hybrid
        combine. Returns (ids, scores, count)."""
        arr = self._refs(refs)
        params = _params(total_documents, n_tokens, threshold, top_k, k)
        out_ids = np.zeros(max(top_k, 1), dtype=np.uint64)
        out_sc = np.zeros(max(top_k, 1), dtype=np.float32)
        out_n = C.c_uint32()
        out_count = C.c_uint64()
        bm_ptr, bm_bits = allow.ffi_args() if allow is not None else (None, 0)
        if vector is None:
            N.check(self._lib.orama_post_search(self._h, arr, len(refs), b, C.byref(params), bm_ptr, bm_bits,
                                                1 if apply_omc else 0, out_ids.ctypes.data, out_sc.ctypes.data,
                                                C.byref(out_n), C.byref(out_count)))
        else:
            if isinstance(vector, dict):
                v_doc, v_sc = _u64(list(vector.keys())), _f32(list(vector.values()))
            else:
                v_doc, v_sc = _u64(vector[0]), _f32(vector[1])
            N.check(self._lib.orama_post_search_hybrid(self._h, arr, len(refs), b, C.byref(params), bm_ptr, bm_bits,
                                                       v_doc.ctypes.data, v_sc.ctypes.data, v_doc.shape[0],
                                                       1 if apply_omc else 0, out_ids.ctypes.data,
                                                       out_sc.ctypes.data, C.byref(out_n), C.byref(out_count)))
        return out_ids[: out_n.value], out_sc[: out_n.value], out_count.value

    def set_avg_len(self, avg_field_len) -> None:
        """Index-wide per-field average lengths for the shards of one index."""
        avg = _f32(avg_field_len)
        N.check(self._lib.orama_post_set_avg_len(self._h, avg.ctypes.data, avg.shape[0]))

    def search_batch(self, queries, total_documents: float, top_k: int, allow: AllowBitmap | None = None,
                     apply_omc: bool = True, max_parallel: int = 8, b: float = B_DEFAULT, k: float = K1_DEFAULT,
                     statuses: bool = False):
        """queries: list of (refs, n_tokens, threshold | None[, top_k of this query <= top_k]).  One call
        (orama_post_search_batch): queries are scored 32 at a time by the range-partitioned scorer; those it does not
        take run on `max_parallel` library threads.  Returns a list of (ids, scores, count), each identical to `search`
        of that query.  With `statuses` (orama_post_search_batch_status) nothing is raised for a failing query: the
        result is (list, status array) and a failed query's entry is empty."""
        prep = self.prepare_batch(queries, total_documents, top_k, allow, apply_omc, max_parallel, b, k)
        st = prep.run(statuses)
        return (prep.results(), st) if statuses else prep.results()

    def prepare_batch(self, queries, total_documents: float, top_k: int, allow: AllowBitmap | None = None,
                      apply_omc: bool = True, max_parallel: int = 8, b: float = B_DEFAULT, k: float = K1_DEFAULT) -> "PreparedBatch":
        """The descriptor array of `search_batch` built once (what a native caller holds anyway): `.run()` is the bare
        orama_post_search_batch call, `.results()` the list `search_batch` returns."""
        return PreparedBatch(self, queries, total_documents, top_k, allow, apply_omc, max_parallel, b, k)

    def search_scores(self, refs, n_tokens: int, total_documents: float, top_k: int, threshold=None,
                      allow: AllowBitmap | None = None, apply_omc: bool = True, b: float = B_DEFAULT,
                      k: float = K1_DEFAULT, vector: dict | None = None) -> ScoreMap:
        """`search` that also keeps the whole score map resident (orama_post_search_scores): returns a ScoreMap whose
        `.hits` are (ids, scores, count)."""
        arr = self._refs(refs)
        params = _params(total_documents, n_tokens, threshold, top_k, k)
        out_ids = np.zeros(max(top_k, 1), dtype=np.uint64)
        out_sc = np.zeros(max(top_k, 1), dtype=np.float32)
        out_n, out_count, h = C.c_uint32(), C.c_uint64(), C.c_void_p()
        bm_ptr, bm_bits = allow.ffi_args() if allow is not None else (None, 0)
        hybrid = vector is not None
        vd = _u64(list(vector) if hybrid else [])
        vs = _f32(list(vector.values()) if hybrid else [])
        N.check(self._lib.orama_post_search_scores(self._h, arr, len(refs), b, C.byref(params), bm_ptr, bm_bits,
                                                   1 if hybrid else 0, vd.ctypes.data, vs.ctypes.data, len(vd),
                                                   1 if apply_omc else 0, out_ids.ctypes.data, out_sc.ctypes.data,
                                                   C.byref(out_n), C.byref(out_count), C.byref(h)))
        return ScoreMap(self._lib, h, (out_ids[: out_n.value].copy(), out_sc[: out_n.value].copy(), out_count.value), self)

    def staged_query(self, refs, n_tokens: int, total_documents: float, top_k: int, d_df_ptr: int, stream: int,
                     threshold=None, allow: AllowBitmap | None = None, apply_omc: bool = True, hybrid: bool = False,
                     n_vec_cap: int = 0, b: float = B_DEFAULT, k: float = K1_DEFAULT) -> "StagedQuery":
        """Stage 1 of the sharded form (orama_post_query_begin): K3 accumulate on this shard, local df per token
        written to the device buffer at `d_df_ptr` (int32[n_tokens]) on HIP stream `stream`."""
        arr = self._refs(refs)
        params = _params(total_documents, n_tokens, threshold, top_k, k)
        bm_ptr, bm_bits = allow.ffi_args() if allow is not None else (None, 0)
        h = C.c_void_p()
        N.check(self._lib.orama_post_query_begin(self._h, arr, len(refs), b, C.byref(params), bm_ptr, bm_bits,
                                                 1 if hybrid else 0, 1 if apply_omc else 0, int(n_vec_cap),
                                                 C.c_void_p(stream), C.c_void_p(d_df_ptr), C.byref(h)))
        sq = StagedQuery(self._lib, h, top_k)
        self._adopt(sq)
        return sq


class PreparedHybrid:
    """One orama_hybrid_search call with its arguments already in C form (PostingsStore.prepare_hybrid)."""

    def __init__(self, post, vec_store, query, limit, similarity, refs, n_tokens, total_documents, top_k, threshold, allow,
                 apply_omc, rescale_e5, b, k):
        self._post, self._vec, self._allow = post, vec_store, allow  # (kept alive)
        self._arr = post._refs(refs)
        self._params = _params(total_documents, n_tokens, threshold, top_k, k)
        self._qv = _f32(query)
        self._out_ids = np.zeros(max(top_k, 1), dtype=np.uint64)
        self._out_sc = np.zeros(max(top_k, 1), dtype=np.float32)
        self._out_n, self._out_count = C.c_uint32(), C.c_uint64()
        bm_ptr, bm_bits = allow.ffi_args() if allow is not None else (None, 0)
        self._args = (vec_store.handle, post._h, self._qv.ctypes.data, int(limit), float(similarity), 1 if rescale_e5 else 0,
                      self._arr, len(refs), b, C.byref(self._params), bm_ptr, bm_bits, 1 if apply_omc else 0,
                      self._out_ids.ctypes.data, self._out_sc.ctypes.data, C.byref(self._out_n), C.byref(self._out_count))
        self._call = post._lib.orama_hybrid_search

    def run(self):
        N.check(self._call(*self._args))
        n = self._out_n.value
        return self._out_ids[:n].copy(), self._out_sc[:n].copy(), self._out_count.value


def post_block_bytes(top_k: int) -> int:
    """Mirror of orama_post_block_bytes: [k u64 ids][k f32 scores][pad to 8][u64 count]."""
    return ((top_k * 12 + 7) & ~7) + 8


class StagedQuery:
    """One query of a sharded index between its stages (include/orama_hip.h, "one index sharded over several
    GPUs").  All calls must come from the thread that created it; `end()` waits for the stream."""

    def __init__(self, lib, handle, top_k: int):
        self._lib = lib
        self._h = handle
        self.top_k = top_k

    def score(self, df_global, d_minmax_ptr: int | None) -> None:
        df = np.ascontiguousarray(df_global, dtype=np.uint32)
        N.check(self._lib.orama_post_query_score(self._h, df.ctypes.data,
                                                 C.c_void_p(d_minmax_ptr) if d_minmax_ptr else None))

    def finish(self, d_minmax_ptr: int | None, vector, d_block_ptr: int) -> None:
        if vector is None:
            v_doc, v_sc = _u64([]), _f32([])
        elif isinstance(vector, dict):
            v_doc, v_sc = _u64(list(vector.keys())), _f32(list(vector.values()))
        else:
            v_doc, v_sc = _u64(vector[0]), _f32(vector[1])
        N.check(self._lib.orama_post_query_finish(self._h, C.c_void_p(d_minmax_ptr) if d_minmax_ptr else None,
                                                  v_doc.ctypes.data, v_sc.ctypes.data, v_doc.shape[0],
                                                  C.c_void_p(d_block_ptr)))

    def end(self) -> None:
        if self._h:
            self._lib.orama_post_query_end(self._h)
            self._h = None

    close = end

    def __del__(self):
        try:
            self.end()
        except Exception:  # noqa: BLE001
            pass


class TermDictionary:
    """The sorted terms of one string field, resident in HBM (orama_dict_*): `expand` is the
    dictionary step of collect_contributions — exact term, or every term the token prefixes plus, with `tolerance`,
    every term within that Levenshtein distance (bytes) — as one device scan over all terms."""

    def __init__(self, ctx: Context, terms):
        self._lib = N.load()
        self.terms = list(terms)
        enc = [t.encode("utf-8") for t in self.terms]
        if any(a >= b for a, b in zip(enc, enc[1:])):
            raise ValueError("terms must be strictly ascending (byte order)")
        blob = np.frombuffer(b"".join(enc), dtype=np.uint8) if enc else np.zeros(0, np.uint8)
        off = np.zeros(len(enc) + 1, dtype=np.uint32)
        if enc:
            off[1:] = np.cumsum([len(e) for e in enc])
        blob = np.ascontiguousarray(blob)
        h = C.c_void_p()
        N.check(self._lib.orama_dict_create(ctx.handle, blob.ctypes.data if blob.size else None, off.ctypes.data,
                                            len(enc), C.byref(h)))
        self._h = h
        ctx._adopt(self)

    def expand(self, token: str, exact: bool = False, tolerance: int = 0) -> list[int]:
        """Indexes (ascending) of the matching terms."""
        tok = np.frombuffer(token.encode("utf-8"), dtype=np.uint8)
        tok = np.ascontiguousarray(tok)
        cap = 1024
        while True:
            out = np.zeros(cap, dtype=np.uint32)
            n = C.c_uint32()
            N.check(self._lib.orama_dict_expand(self._h, tok.ctypes.data if tok.size else None, tok.size,
                                                1 if exact else 0, int(tolerance or 0), cap, out.ctypes.data,
                                                C.byref(n)))
            if n.value <= cap:
                return out[: n.value].tolist()
            cap = int(n.value)

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._lib.orama_dict_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


class PostSearchBatcher:
    """Request batcher in front of one PostingsStore (orama_post_batcher_*): concurrent single-query callers (threads)
    are scored 32 at a time by the range-partitioned scorer.  `search` has PostingsStore.search's arguments and
    results (non-hybrid)."""

    def __init__(self, store: PostingsStore, max_batch: int = 256, max_wait_us: int = 0, group=None, shards=None):
        """group + shards: the batcher in front of a shard group (orama_post_batcher_create_group); `store` is then any one
        of the shards (it marshals the references)."""
        self._lib = N.load()
        self.store = store
        h = C.c_void_p()
        if group is not None:
            self._group, self._shards = group, list(shards)
            N.check(self._lib.orama_post_batcher_create_group(group._h, group._handles(shards), int(max_batch), int(max_wait_us),
                                                              C.byref(h)))
        else:
            N.check(self._lib.orama_post_batcher_create(store._h, int(max_batch), int(max_wait_us), C.byref(h)))
        self._h = h
        store._adopt(self)

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._lib.orama_post_batcher_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def search(self, refs, n_tokens: int, total_documents: float, top_k: int, threshold=None,
               allow: AllowBitmap | None = None, apply_omc: bool = True, b: float = B_DEFAULT, k: float = K1_DEFAULT):
        arr = self.store._refs(refs)
        params = _params(total_documents, n_tokens, threshold, top_k, k)
        out_ids = np.zeros(max(top_k, 1), dtype=np.uint64)
        out_sc = np.zeros(max(top_k, 1), dtype=np.float32)
        out_n, out_count = C.c_uint32(), C.c_uint64()
        bm_ptr, bm_bits = allow.ffi_args() if allow is not None else (None, 0)
        N.check(self._lib.orama_post_batcher_search(self._h, arr, len(refs), b, C.byref(params), bm_ptr, bm_bits,
                                                    1 if apply_omc else 0, out_ids.ctypes.data, out_sc.ctypes.data,
                                                    C.byref(out_n), C.byref(out_count)))
        return out_ids[: out_n.value], out_sc[: out_n.value], out_count.value

    def stats(self) -> dict:
        r, bt, l = C.c_uint64(), C.c_uint64(), C.c_uint32()
        N.check(self._lib.orama_post_batcher_stats(self._h, C.byref(r), C.byref(bt), C.byref(l)))
        return {"requests": r.value, "batches": bt.value, "largest_batch": l.value,
                "mean_batch": (r.value / bt.value) if bt.value else 0.0}
