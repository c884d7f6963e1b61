"""K3r — the document-range partitioned BM25F scorer (bm25_ranges.hip) — against the oracle and against K3 (the
per-document-record scorer), bit for bit: filters, several lists per token (df counted on the device), thresholds,
OMC, more than 32 tokens, documents clustered in id space (range overflow -> smaller ranges), queries the sort key
cannot hold (fall back to K3), and the batch entry with mixed queries."""
import numpy as np
import pytest

import oramacore_amd as oa
from oramacore_amd import fulltext as ft
from oracle import oracle as orc  # checker only

pytestmark = pytest.mark.gpu
F = np.float32


@pytest.fixture(scope="module")
def ctx():
    c = oa.Context(0)
    yield c
    c.set_bm25_ranges(True)
    c.close()


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def ntf_of(tf, length, avg, boost, b=0.75):
    """bm25f normalised tf in f32, one rounding per operation (what the resident path computes on the device)."""
    tf, length = np.asarray(tf, dtype=F), np.asarray(length, dtype=F)
    return (F(boost) * (tf / ((F(1.0) - F(b)) + F(b) * (length / F(avg))))).astype(F)


def test_ntf_helper_matches_the_oracle_function():
    rng = np.random.default_rng(0)
    for _ in range(200):
        tf, ln, avg = int(rng.integers(1, 50)), int(rng.integers(1, 3000)), float(F(rng.uniform(1, 500)))
        assert bits([ntf_of([tf], [ln], avg, 1.0)[0]])[0] == bits([orc.bm25f_normalized_tf(tf, ln, avg, 0.75)])[0]


class Corpus:
    """Random resident store: `n_fields` fields, explicit lists; remembers what the oracle needs."""

    def __init__(self, ctx, n_docs, lists, avg, doc_ids=None, seed=0):
        rng = np.random.default_rng(seed)
        self.n_docs = n_docs
        self.doc_ids = np.arange(n_docs, dtype=np.uint64) if doc_ids is None else doc_ids
        self.avg = [float(F(a)) for a in avg]
        self.lists = []
        pls = []
        for field, local in lists:  # local: sorted unique local doc indices
            local = np.asarray(local, dtype=np.int64)
            tf = rng.integers(1, 6, size=len(local))
            ln = rng.integers(1, 400, size=len(local))
            self.lists.append((field, local, tf, ln))
            pls.append(ft.PostingList(field=field, docs=self.doc_ids[local], tf=tf, field_len=ln))
        self.store = ft.PostingsStore(ctx)
        self.store.build(self.doc_ids, self.avg, pls)

    def entries(self, refs, allow_mask=None):
        out = []
        for tok, lst, boost in refs:
            field, local, tf, ln = self.lists[lst]
            keep = np.ones(len(local), dtype=bool) if allow_mask is None else allow_mask[local]
            out.append((tok, self.doc_ids[local[keep]], ntf_of(tf[keep], ln[keep], self.avg[field], boost)))
        return out

    def oracle(self, refs, n_tok, top_k, thr=None, allow_mask=None, omc=None):
        docs, scores = orc.search_full_text(self.entries(refs, allow_mask), n_tok, float(self.n_docs), 1.2, thr)
        if omc:
            scores = orc.apply_omc(docs, scores, list(omc), list(omc.values()))
        td, ts = orc.top_n(docs, scores, top_k)
        return td, ts, len(docs)


def check(ctx, corpus, refs, n_tok, top_k, thr=None, allow=None, allow_mask=None, omc=None, tag=""):
    od, os_, ocount = corpus.oracle(refs, n_tok, top_k, thr, allow_mask, omc)
    got = {}
    for ranges in (True, False):
        ctx.set_bm25_ranges(ranges)
        ids, sc, count = corpus.store.search(refs, n_tok, float(corpus.n_docs), top_k, thr, allow=allow,
                                             apply_omc=omc is not None)
        assert count == ocount, (tag, ranges, count, ocount)
        assert ids.tolist() == od.tolist(), (tag, ranges)
        assert np.array_equal(bits(sc), bits(os_)), (tag, ranges)
        got[ranges] = (ids, sc)
    ctx.set_bm25_ranges(True)


def random_lists(rng, n_docs, n_lists, n_fields, lo, hi):
    out = []
    for l in range(n_lists):
        n = int(rng.integers(lo, min(hi, n_docs) + 1))
        out.append((l % n_fields, np.sort(rng.choice(n_docs, size=n, replace=False))))
    return out


def test_ranges_equal_oracle_and_k3(ctx):
    rng = np.random.default_rng(1)
    n_docs = 40_000
    doc_ids = np.cumsum(rng.integers(1, 4, size=n_docs)).astype(np.uint64) + np.uint64(10**9)  # non-dense 64-bit ids
    corpus = Corpus(ctx, n_docs, random_lists(rng, n_docs, 24, 3, 50, 9000), [120.0, 33.5, 7.25], doc_ids, seed=2)
    allow_mask = rng.random(n_docs) < 0.6
    bm = oa.AllowBitmap(int(doc_ids.max()) + 1, doc_ids[allow_mask])
    corpus.store.set_omc({int(doc_ids[i]): float(m) for i, m in zip(rng.choice(n_docs, 300, replace=False),
                                                                      rng.choice([0.25, 0.5, 2.0, 5.0], 300))})
    omc = None
    for case in range(12):
        n_tok = int(rng.integers(1, 9))
        refs = []
        for t in range(n_tok):
            for l in rng.choice(24, size=int(rng.integers(1, 4)), replace=False):  # 1-3 lists per token: df on the device
                refs.append((t, int(l), float(F(rng.choice([1.0, 0.5, 2.0, 3.5])))))
        rng.shuffle(refs)  # reference order is the accumulation order inside a token, whatever the interleaving
        thr = None if case % 3 else int(rng.integers(1, n_tok + 1))
        for filt in (False, True):
            check(ctx, corpus, refs, n_tok, int(rng.choice([1, 10, 100, 1000])), thr, bm if filt else None,
                  allow_mask if filt else None, omc, tag=("mixed", case, filt))
    # one list per token and no filter: df known on the host, a single scoring pass
    for case in range(4):
        n_tok = int(rng.integers(1, 13))
        refs = [(t, int(l), 1.0) for t, l in enumerate(rng.choice(24, size=n_tok, replace=False))]
        check(ctx, corpus, refs, n_tok, 50, None, tag=("single", case))
    # the dense multipliers of the store
    omc_map = {}
    ids0, _, _ = corpus.store.search([(0, 0, 1.0)], 1, float(n_docs), 20, apply_omc=False)
    omc_map = {int(ids0[0]): 0.001, int(ids0[1]): 7.0}
    corpus.store.set_omc(omc_map)
    check(ctx, corpus, [(0, 0, 1.0), (1, 5, 2.0)], 2, 30, None, omc=omc_map, tag="omc")
    corpus.store.close()


def test_many_tokens_wrap_the_threshold_mask(ctx):
    rng = np.random.default_rng(3)
    n_docs = 3000
    corpus = Corpus(ctx, n_docs, random_lists(rng, n_docs, 64, 2, 100, 1500), [50.0, 9.0], seed=4)
    refs = [(t, t, 1.0) for t in range(64)]
    for thr in (None, 5, 20, 33):
        check(ctx, corpus, refs, 64, 200, thr, tag=("64 tokens", thr))
    # 33 .. 64 lists take the range scorer's 64-bit presence masks (one bit per list); more than 64 lists fall to the
    # per-record scorer — same answers either way
    refs = [(t, (3 * t + j) % 64, 1.0 + 0.5 * j) for t in range(20) for j in range(2)]   # 40 lists, two per token
    check(ctx, corpus, refs, 20, 150, 3, tag="40 lists over 20 tokens")
    refs = [(t, (5 * t + j) % 64, 1.0) for t in range(24) for j in range(3)]             # 72 lists: beyond the masks
    check(ctx, corpus, refs, 24, 150, None, tag="72 lists over 24 tokens")
    corpus.store.close()


def test_clustered_documents_shrink_the_ranges(ctx):
    """5 000 postings of one term inside 5 000 consecutive documents of a 1 M-document index: the first choice of
    range width (tens of thousands of documents) overflows a workgroup's 2 048 slots and the query reruns with 8x
    and 64x smaller ranges; another term is everywhere (one posting per document: a range never exceeds its width)."""
    rng = np.random.default_rng(5)
    n_docs = 1_000_000
    cluster = np.arange(400_000, 405_000)
    sparse = np.sort(rng.choice(n_docs, size=300, replace=False))
    dense_run = np.arange(0, 6000)
    corpus = Corpus(ctx, n_docs, [(0, cluster), (0, sparse), (1, dense_run), (1, cluster[::3])], [40.0, 12.0], seed=6)
    check(ctx, corpus, [(0, 0, 1.0), (1, 1, 1.0)], 2, 100, tag="cluster")
    check(ctx, corpus, [(0, 0, 1.0), (0, 3, 2.0), (1, 2, 1.0)], 2, 1000, 2, tag="cluster, two lists per token")
    res = corpus.store.search_batch([([(0, 0, 1.0)], 1, None), ([(0, 1, 1.0)], 1, None), ([(0, 2, 1.0), (1, 0, 1.0)], 2, 1)],
                                    float(n_docs), 64)
    for (refs, n_tok, thr), (ids, sc, count) in zip([([(0, 0, 1.0)], 1, None), ([(0, 1, 1.0)], 1, None),
                                                     ([(0, 2, 1.0), (1, 0, 1.0)], 2, 1)], res):
        od, os_, ocount = corpus.oracle(refs, n_tok, 64, thr)
        assert count == ocount and ids.tolist() == od.tolist() and np.array_equal(bits(sc), bits(os_))
    corpus.store.close()


def test_terms_that_occur_together_overflow_the_cell_tables(ctx):
    """Round 4's scoring launch keeps the documents with more than one posting of a range in bounded LDS tables (1 024 cells,
    512 documents).  Terms that occur in the SAME documents — every document of a range is then a multi-posting document —
    overflow them at the first choice of range width: the query must rerun with narrower ranges and still return the
    oracle's bits.  Also: three terms over the same documents with a threshold, two fields of one token (several lists per
    token: the cells are summed in reference order), and a mix of both with a filter."""
    rng = np.random.default_rng(15)
    n_docs = 400_000
    together = np.sort(rng.choice(n_docs, size=90_000, replace=False))
    half = together[::2]
    other = np.sort(rng.choice(n_docs, size=60_000, replace=False))
    corpus = Corpus(ctx, n_docs, [(0, together), (0, together), (0, together), (1, together), (0, half), (1, other)], [30.0, 9.0], seed=16)
    check(ctx, corpus, [(0, 0, 1.0), (1, 1, 1.0)], 2, 100, tag="two terms, same documents")
    check(ctx, corpus, [(0, 0, 1.0), (1, 1, 2.0), (2, 2, 0.5)], 3, 250, 3, tag="three terms, same documents, threshold")
    check(ctx, corpus, [(0, 0, 1.0), (0, 3, 1.5), (1, 4, 1.0)], 2, 100, tag="two fields of one token + a subset term")
    mask = (np.arange(n_docs) % 5) != 2
    check(ctx, corpus, [(0, 0, 1.0), (0, 3, 1.5), (1, 1, 1.0), (1, 5, 1.0), (2, 4, 1.0)], 3, 64, 2, allow=oa.AllowBitmap.from_mask(mask),
          allow_mask=mask, tag="mixed, filtered")
    corpus.store.close()


def test_batch_with_mixed_queries(ctx):
    """orama_post_search_batch: queries with different token counts, thresholds and top_k (0 = count only), an empty
    query, a query whose 1 100 references do not fit the sort key (falls back to K3), all under one filter."""
    rng = np.random.default_rng(7)
    n_docs = 20_000
    lists = random_lists(rng, n_docs, 40, 2, 20, 3000) + [(0, np.array([], dtype=np.int64))]
    corpus = Corpus(ctx, n_docs, lists, [80.0, 15.0], seed=8)
    allow_mask = rng.random(n_docs) < 0.8
    bm = oa.AllowBitmap(n_docs, np.nonzero(allow_mask)[0].astype(np.uint64))
    queries = []
    for i in range(70):  # > 2 chunks of 32
        n_tok = int(rng.integers(1, 7))
        refs = [(t, int(rng.integers(0, 40)), 1.0) for t in range(n_tok) for _ in range(int(rng.integers(1, 3)))]
        queries.append((refs, n_tok, None if i % 4 else 1, int(rng.choice([0, 1, 10, 64]))))
    queries.append(([], 1, None, 10))                                   # no references
    queries.append(([(0, 40, 1.0)], 1, None, 10))                       # only an empty list
    queries.append(([(0, int(l % 40), 1.0) for l in range(1100)], 1, None, 25))  # too many lists for the sort key
    for allow, mask in ((None, None), (bm, allow_mask)):
        res = corpus.store.search_batch(queries, float(n_docs), 64, allow=allow)
        assert len(res) == len(queries)
        for (refs, n_tok, thr, k), (ids, sc, count) in zip(queries, res):
            od, os_, ocount = corpus.oracle(refs, n_tok, k, thr, mask)
            assert count == ocount, (len(refs), k)
            assert ids.tolist() == od.tolist() and np.array_equal(bits(sc), bits(os_)), (len(refs), k)
    corpus.store.close()


def test_appended_lists_and_dense_ids(ctx):
    """Delta lists appended after the build (orama_post_append) are sorted like built ones: the range scorer reads
    both; dense ids take the implicit doc table."""
    rng = np.random.default_rng(9)
    n0, n1 = 5000, 2000
    corpus = Corpus(ctx, n0, random_lists(rng, n0, 6, 1, 200, 2500), [60.0], seed=10)
    new_docs = np.arange(n0, n0 + n1, dtype=np.uint64)
    delta = []
    for l in range(3):
        local = np.sort(rng.choice(n0 + n1, size=900, replace=False))
        local = local[local >= n0] if l == 0 else local  # one list of new documents only
        tf, ln = rng.integers(1, 6, size=len(local)), rng.integers(1, 400, size=len(local))
        corpus.lists.append((0, local, tf, ln))
        delta.append(ft.PostingList(field=0, docs=local.astype(np.uint64), tf=tf, field_len=ln))
    corpus.doc_ids = np.arange(n0 + n1, dtype=np.uint64)
    corpus.n_docs = n0 + n1
    corpus.avg = [float(F(61.5))]
    first = corpus.store.append(new_docs, corpus.avg, delta)
    assert first == 6
    check(ctx, corpus, [(0, 0, 1.0), (0, 6, 1.0), (1, 7, 2.0), (2, 8, 1.0), (2, 3, 1.0)], 3, 500, tag="append")
    corpus.store.close()


def test_the_plain_instantiation_equals_the_general_one(ctx):
    """The scoring launch has an instantiation of its own for batches without score map, OMC multipliers and min / max (singleton
    scores evaluated by every lane and selected, not branched to).  The same queries through the general form — forced by an OMC
    table whose only multiplier is 1.0 — must give the same bits: plain, threshold, several lists per token, filtered, k = 1..300."""
    rng = np.random.default_rng(123)
    n_docs = 80_000
    corpus = Corpus(ctx, n_docs, random_lists(rng, n_docs, 10, 2, 3000, 30000), [40.0, 12.0], seed=124)
    mask = rng.random(n_docs) < 0.7
    bm = oa.AllowBitmap(n_docs, np.nonzero(mask)[0].astype(np.uint64))
    ctx.set_bm25_ranges(True)
    cases = [([(0, 0, 1.0), (1, 1, 1.0), (2, 2, 1.5)], 3, 100, None, None),
             ([(0, 3, 1.0), (1, 4, 1.0), (2, 5, 1.0), (3, 6, 2.0)], 4, 10, 2, None),
             ([(0, 0, 1.0), (0, 1, 2.0), (1, 7, 1.0), (1, 8, 1.0), (1, 9, 0.5)], 2, 300, None, None),
             ([(0, 2, 1.0), (1, 3, 1.0)], 2, 1, None, bm),
             ([(0, 5, 1.0), (1, 6, 1.0), (2, 7, 1.0)], 3, 64, 3, bm)]
    plain = [corpus.store.search(r, nt, float(n_docs), k, thr, allow=al, apply_omc=False) for r, nt, k, thr, al in cases]
    corpus.store.set_omc({int(n_docs - 1): 1.0})  # a multiplier table exists: the general form runs
    general = [corpus.store.search(r, nt, float(n_docs), k, thr, allow=al, apply_omc=True) for r, nt, k, thr, al in cases]
    corpus.store.set_omc({})
    for i, ((pi, ps, pc), (gi, gs, gc)) in enumerate(zip(plain, general)):
        assert pc == gc and pi.tolist() == gi.tolist() and np.array_equal(bits(ps), bits(gs)), i
        r, nt, k, thr, al = cases[i]
        od, os_, ocount = corpus.oracle(r, nt, k, thr, mask if al is not None else None)
        assert pc == ocount and pi.tolist() == od.tolist() and np.array_equal(bits(ps), bits(os_)), i
    corpus.store.close()


def test_unions_of_a_tokens_lists_are_counted_once_per_index(ctx):
    """A token with several lists (one per field, expansions) needs |union of the lists| as its document frequency: counted on
    the device by the first query that brings the set (a second scoring-sized launch), remembered under the list ids until
    the postings change.  Same answers — the oracle's — with and without the remembered counts; a filtered query never uses
    them (its count depends on the filter); an append forgets them."""
    rng = np.random.default_rng(77)
    n0 = 30_000
    corpus = Corpus(ctx, n0, random_lists(rng, n0, 8, 2, 2000, 9000), [55.0, 9.0], seed=78)
    refs = [(0, 0, 1.0), (0, 1, 2.0), (1, 2, 1.0), (2, 3, 1.0), (2, 4, 1.5), (2, 5, 1.0)]
    refs_same_sets = [(0, 1, 1.0), (0, 0, 1.0), (1, 6, 1.0), (2, 5, 1.0), (2, 3, 2.0), (2, 4, 1.0)]  # other order, other boosts

    def df_launches(fn):
        ctx.prof_reset()
        ctx.prof_enable(True)
        fn()
        ctx.prof_enable(False)
        return ctx.prof_get("bm25_range_df")[1]

    ctx.set_bm25_ranges(True)
    od, os_, ocount = corpus.oracle(refs, 3, 40)
    for attempt in range(3):
        got = {}
        n_df = df_launches(lambda: got.update(r=corpus.store.search(refs, 3, float(n0), 40)))
        ids, sc, count = got["r"]
        assert count == ocount and ids.tolist() == od.tolist() and np.array_equal(bits(sc), bits(os_)), attempt
        assert (n_df > 0) == (attempt == 0), (attempt, n_df)  # counted by the first query only
    od2, os2, oc2 = corpus.oracle(refs_same_sets, 3, 40)
    got = {}
    assert df_launches(lambda: got.update(r=corpus.store.search(refs_same_sets, 3, float(n0), 40))) == 0  # the same sets of lists
    assert got["r"][2] == oc2 and got["r"][0].tolist() == od2.tolist() and np.array_equal(bits(got["r"][1]), bits(os2))
    # a batch: queries with remembered sets beside one with a new set
    new_set = [(0, 0, 1.0), (0, 7, 1.0), (1, 2, 1.0)]
    res = corpus.store.search_batch([(refs, 3, None, 40), (new_set, 2, None, 25), (refs_same_sets, 3, None, 40)], float(n0), 64)
    for (ids, sc, count), (r, nt, k) in zip(res, ((refs, 3, 40), (new_set, 2, 25), (refs_same_sets, 3, 40))):
        xd, xs, xc = corpus.oracle(r, nt, k)
        assert count == xc and ids.tolist() == xd.tolist() and np.array_equal(bits(sc), bits(xs))
    # under a filter the count is the filter's
    mask = rng.random(n0) < 0.5
    bm = oa.AllowBitmap(n0, np.nonzero(mask)[0].astype(np.uint64))
    fd, fs, fc = corpus.oracle(refs, 3, 40, None, mask)
    got = {}
    assert df_launches(lambda: got.update(r=corpus.store.search(refs, 3, float(n0), 40, allow=bm))) > 0
    assert got["r"][2] == fc and got["r"][0].tolist() == fd.tolist() and np.array_equal(bits(got["r"][1]), bits(fs))
    # ... unless the filter is a RESIDENT bitmap (the NOT-deleted bitmap of an index with pending deletes): remembered under
    # the version of its content — until somebody sets a bit
    res_bm = bm.to_device(ctx)
    single = [(0, 0, 1.0), (1, 2, 1.0), (2, 4, 1.0)]  # one list per token: under a filter their counts are the filter's too
    for r, nt in ((refs, 3), (single, 3)):
        xd, xs, xc = corpus.oracle(r, nt, 40, None, mask)
        for attempt in range(2):
            got = {}
            n_df = df_launches(lambda: got.update(r=corpus.store.search(r, nt, float(n0), 40, allow=res_bm)))
            assert got["r"][2] == xc and got["r"][0].tolist() == xd.tolist() and np.array_equal(bits(got["r"][1]), bits(xs)), attempt
            assert (n_df > 0) == (attempt == 0), (attempt, n_df)
    gone = np.nonzero(mask)[0][:700].astype(np.uint64)
    res_bm.set(gone, False)
    mask2 = mask.copy()
    mask2[gone.astype(np.int64)] = False
    xd, xs, xc = corpus.oracle(refs, 3, 40, None, mask2)
    got = {}
    assert df_launches(lambda: got.update(r=corpus.store.search(refs, 3, float(n0), 40, allow=res_bm))) > 0  # new content: counted again
    assert got["r"][2] == xc and got["r"][0].tolist() == xd.tolist() and np.array_equal(bits(got["r"][1]), bits(xs))
    res_bm.close()
    # an append changes the lists' neighbours and the averages: nothing remembered survives it
    n1 = 500
    new_docs = np.arange(n0, n0 + n1, dtype=np.uint64)
    local = np.sort(rng.choice(n1, size=300, replace=False)) + n0
    tf, ln = rng.integers(1, 6, size=len(local)), rng.integers(1, 400, size=len(local))
    corpus.lists.append((0, local, tf, ln))
    corpus.doc_ids = np.arange(n0 + n1, dtype=np.uint64)
    corpus.n_docs = n0 + n1
    corpus.store.append(new_docs, corpus.avg, [ft.PostingList(field=0, docs=local.astype(np.uint64), tf=tf, field_len=ln)])
    od3, os3, oc3 = corpus.oracle(refs, 3, 40)
    got = {}
    assert df_launches(lambda: got.update(r=corpus.store.search(refs, 3, float(n0 + n1), 40))) > 0
    assert got["r"][2] == oc3 and got["r"][0].tolist() == od3.tolist() and np.array_equal(bits(got["r"][1]), bits(os3))
    corpus.store.close()


def test_request_batcher_equals_direct_searches(ctx):
    """orama_post_batcher_*: 16 threads of single-query requests (two filters, different k, thresholds) are coalesced
    into batches; every answer equals the direct orama_post_search of the same request and the oracle's."""
    import threading

    rng = np.random.default_rng(11)
    n_docs = 30_000
    corpus = Corpus(ctx, n_docs, random_lists(rng, n_docs, 30, 2, 50, 4000), [70.0, 11.0], seed=12)
    allow_mask = rng.random(n_docs) < 0.7
    bm = oa.AllowBitmap(n_docs, np.nonzero(allow_mask)[0].astype(np.uint64))
    reqs = []
    for i in range(160):
        n_tok = int(rng.integers(1, 6))
        refs = [(t, int(rng.integers(0, 30)), 1.0) for t in range(n_tok) for _ in range(int(rng.integers(1, 3)))]
        reqs.append((refs, n_tok, None if i % 3 else 1, int(rng.choice([1, 10, 50])), i % 2 == 1))
    batcher = ft.PostSearchBatcher(corpus.store, max_batch=64)
    got = [None] * len(reqs)
    errors = []

    def worker(t):
        try:
            for i in range(t, len(reqs), 16):
                refs, n_tok, thr, k, filt = reqs[i]
                got[i] = batcher.search(refs, n_tok, float(n_docs), k, thr, allow=bm if filt else None)
        except Exception as e:  # noqa: BLE001
            errors.append(e)

    ths = [threading.Thread(target=worker, args=(t,)) for t in range(16)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not errors, errors
    for (refs, n_tok, thr, k, filt), (ids, sc, count) in zip(reqs, got):
        od, os_, ocount = corpus.oracle(refs, n_tok, k, thr, allow_mask if filt else None)
        assert count == ocount and ids.tolist() == od.tolist() and np.array_equal(bits(sc), bits(os_))
        d_ids, d_sc, d_count = corpus.store.search(refs, n_tok, float(n_docs), k, thr, allow=bm if filt else None)
        assert d_count == count and d_ids.tolist() == ids.tolist() and np.array_equal(bits(d_sc), bits(sc))
    st = batcher.stats()
    assert st["requests"] == len(reqs) and st["batches"] <= len(reqs)
    # a malformed request fails alone
    with pytest.raises(oa.OramaError):
        batcher.search([(3, 0, 1.0)], 2, float(n_docs), 10)
    ids, sc, count = batcher.search([(0, 0, 1.0)], 1, float(n_docs), 5)
    assert len(ids) == 5
    batcher.close()
    corpus.store.close()


def test_hybrid_on_the_range_scorer(ctx):
    """orama_post_search_hybrid on K3r (no OMC): candidates + exact scores of the vector hits + host combine — against
    the oracle's normalize_and_combine + top_n and against K3, bit for bit; cases that must fall back (ties at the cut,
    OMC, empty full-text side) still answer correctly."""
    rng = np.random.default_rng(31)
    n_docs = 50_000
    corpus = Corpus(ctx, n_docs, random_lists(rng, n_docs, 20, 2, 100, 8000), [90.0, 14.0], seed=32)
    allow_mask = rng.random(n_docs) < 0.7
    bm = oa.AllowBitmap(n_docs, np.nonzero(allow_mask)[0].astype(np.uint64))

    def oracle(refs, n_tok, k, vec, thr=None, mask=None, omc=None):
        fd, fs = orc.search_full_text(corpus.entries(refs, mask), n_tok, float(n_docs), 1.2, thr)
        od, os_ = orc.normalize_and_combine(list(vec), list(vec.values()), fd, fs)
        if omc:
            os_ = orc.apply_omc(od, os_, list(omc), list(omc.values()))
        td, ts = orc.top_n(od, os_, k)
        return td, ts, len(od)

    ctx.prof_reset()
    ctx.prof_enable(True)
    for case in range(10):
        n_tok = int(rng.integers(1, 6))
        refs = [(t, int(l), float(F(rng.choice([1.0, 2.0])))) for t in range(n_tok)
                for l in rng.choice(20, size=int(rng.integers(1, 3)), replace=False)]
        fd, fs = orc.search_full_text(corpus.entries(refs), n_tok, float(n_docs), 1.2, None)
        inside = rng.choice(fd, size=min(30, len(fd)), replace=False)
        outside = np.setdiff1d(rng.choice(n_docs, size=40, replace=False).astype(np.uint64), fd)[:20]
        vec = {int(d): float(F(s)) for d, s in zip(np.concatenate([inside, outside]), rng.uniform(-0.3, 1.0, size=len(inside) + len(outside)))}
        for k in (1, 10, 100):
            for filt in (False, True):
                thr = None if case % 2 else 1
                od, os_, ocount = oracle(refs, n_tok, k, vec, thr, allow_mask if filt else None)
                for mode in (True, False):
                    ctx.set_bm25_ranges(True, hybrid=mode)
                    ids, sc, count = corpus.store.search(refs, n_tok, float(n_docs), k, thr, allow=bm if filt else None,
                                                         vector=vec, apply_omc=False)
                    assert count == ocount, (case, k, filt, mode)
                    assert ids.tolist() == od.tolist(), (case, k, filt, mode)
                    assert np.array_equal(bits(sc), bits(os_)), (case, k, filt, mode)
    ctx.set_bm25_ranges(True)
    ctx.prof_enable(False)
    assert ctx.prof_get("bm25_range_score")[1] > 0  # the range scorer really served hybrid queries
    # ties at the cut: every document has the same score — the candidate argument cannot decide, K3 answers
    docs = np.arange(3000, dtype=np.int64)
    flat = Corpus(ctx, 3000, [(0, docs)], [7.0], seed=1)
    flat.lists[0] = (0, docs, np.ones(3000, dtype=np.int64), np.full(3000, 7, dtype=np.int64))
    flat.store.build(flat.doc_ids, flat.avg, [ft.PostingList(field=0, docs=flat.doc_ids, tf=np.ones(3000), field_len=np.full(3000, 7))])
    vec = {2999: 0.9, 5: 0.1}
    ids, sc, count = flat.store.search([(0, 0, 1.0)], 1, 3000.0, 10, vector=vec, apply_omc=False)
    fd, fs = orc.search_full_text(flat.entries([(0, 0, 1.0)]), 1, 3000.0, 1.2, None)
    od, os_ = orc.normalize_and_combine(list(vec), list(vec.values()), fd, fs)
    td, ts = orc.top_n(od, os_, 10)
    assert count == len(od) and ids.tolist() == td.tolist() and np.array_equal(bits(sc), bits(ts))
    # OMC applies -> K3; empty full-text side -> K3 (the vector map alone)
    corpus.store.set_omc({int(corpus.doc_ids[7]): 3.0})
    refs = [(0, 0, 1.0), (1, 3, 1.0)]
    vec = {int(corpus.doc_ids[7]): 0.5, int(corpus.doc_ids[11]): 0.7}
    od, os_, ocount = oracle(refs, 2, 20, vec, omc={int(corpus.doc_ids[7]): 3.0})
    ids, sc, count = corpus.store.search(refs, 2, float(n_docs), 20, vector=vec, apply_omc=True)
    assert count == ocount and ids.tolist() == od.tolist() and np.array_equal(bits(sc), bits(os_))
    ids, sc, count = corpus.store.search([], 1, float(n_docs), 5, vector=vec, apply_omc=False)
    od, os_ = orc.normalize_and_combine(list(vec), list(vec.values()), np.zeros(0, np.uint64), np.zeros(0, np.float32))
    td, ts = orc.top_n(od, os_, 5)
    assert count == 2 and ids.tolist() == td.tolist() and np.array_equal(bits(sc), bits(ts))
    flat.store.close()
    corpus.store.close()


def test_a_failing_query_fails_alone(ctx):
    """queries of a batch are independent.  A malformed query (list out of range, token index >= n_tokens,
    top_k above the stride) gets its own status; every other query of the batch is answered exactly as alone."""
    rng = np.random.default_rng(17)
    n_docs = 8_000
    corpus = Corpus(ctx, n_docs, random_lists(rng, n_docs, 12, 2, 50, 2000), [60.0, 12.0], seed=18)
    good = [([(t, int(rng.integers(0, 12)), 1.0) for t in range(3)], 3, None, 10) for _ in range(40)]
    bad = {5: ([(0, 99, 1.0)], 1, None, 10),             # list 99 does not exist
           17: ([(3, 1, 1.0)], 2, None, 10),             # token 3 of a 2-token query
           33: ([(0, 1, 1.0)], 1, None, 65)}             # top_k above the output stride (64)
    queries = [bad.get(i, g) for i, g in enumerate(good)]
    res, st = corpus.store.search_batch(queries, float(n_docs), 64, statuses=True)
    for i, ((refs, n_tok, thr, k), (ids, sc, count)) in enumerate(zip(queries, res)):
        if i in bad:
            assert st[i] == oa._native.ORAMA_ERR_INVALID and len(ids) == 0 and count == 0, i
            continue
        assert st[i] == 0
        od, os_, ocount = corpus.oracle(refs, n_tok, k, thr)
        assert count == ocount and ids.tolist() == od.tolist() and np.array_equal(bits(sc), bits(os_)), i
    # the single-status form reports the first failure and has still answered the rest
    with pytest.raises(oa.OramaError, match="query 5"):
        corpus.store.search_batch(queries, float(n_docs), 64)
    corpus.store.close()


def test_open_score_maps_do_not_starve_searches(ctx):
    """an orama_scores handle keeps its scratch set until it is destroyed; such sets do not count against
    the context's in-flight bound (32), so 40 open handles leave every other call servable."""
    rng = np.random.default_rng(19)
    n_docs = 3_000
    corpus = Corpus(ctx, n_docs, random_lists(rng, n_docs, 6, 1, 50, 900), [40.0], seed=20)
    refs = [(0, 0, 1.0), (1, 3, 1.0)]
    maps = [corpus.store.search_scores(refs, 2, float(n_docs), 5) for _ in range(40)]
    ids, sc, count = corpus.store.search(refs, 2, float(n_docs), 10)       # would wait for ever if handles counted
    od, os_, ocount = corpus.oracle(refs, 2, 10)
    assert count == ocount and ids.tolist() == od.tolist() and np.array_equal(bits(sc), bits(os_))
    assert len(maps[39]) == ocount
    for m in maps:
        m.close()
    corpus.store.close()


def test_one_call_hybrid_search_on_the_range_scorer(ctx):
    """orama_hybrid_search with its full-text leg on K3r and the tail on candidates (raw top-(k + limit + 1) selected
    while the scan runs, the vector hits scored by document afterwards): against the oracle and against the same call
    on K3's per-document records, bit for bit — plain and shadow (two-stage) vector stores, E5 rescale, cut-off,
    filter, several rows per document, k and limit from 1 to 300; ties at the cut and OMC fall back and still answer."""
    import util

    rng = np.random.default_rng(77)
    n_docs, dim = 30_000, 64
    corpus = Corpus(ctx, n_docs, random_lists(rng, n_docs, 16, 2, 200, 6000), [80.0, 12.0], doc_ids=np.arange(n_docs, dtype=np.uint64) * 3 + 5, seed=78)
    rows = util.gaussian_rows(n_docs + 4000, dim, seed=79)
    row_doc = np.concatenate([corpus.doc_ids, corpus.doc_ids[rng.choice(n_docs, size=4000, replace=False)]])  # some documents own two rows
    allow_mask = rng.random(n_docs) < 0.6
    bm = oa.AllowBitmap(int(corpus.doc_ids.max()) + 1, corpus.doc_ids[allow_mask])
    stores = {"plain": oa.EmbeddingFieldStorage(ctx, dimensions=dim), "shadow": oa.EmbeddingFieldStorage(ctx, dimensions=dim, dtype=oa._native.DTYPE_F32_SHADOW16)}
    for st in stores.values():
        st.insert_rows(row_doc, rows)

    served = 0
    for case in range(8):
        n_tok = int(rng.integers(1, 5))
        refs = [(t, int(l), float(F(rng.choice([1.0, 1.5])))) for t in range(n_tok) for l in rng.choice(16, size=int(rng.integers(1, 3)), replace=False)]
        q = (rows[int(rng.integers(len(rows)))] + F(0.4) * rows[int(rng.integers(len(rows)))]).astype(F)
        k, limit = [(10, 10), (1, 300), (100, 100), (300, 1), (25, 40), (100, 7), (5, 64), (50, 200)][case]
        sim, e5, filt = [(0.0, False, False), (0.0, False, True), (0.5, False, False), (0.0, True, False)][case % 4]
        thr = 1 if case == 5 else None
        got = {}
        for name, st in stores.items():
            for mode in (True, False):
                ctx.set_bm25_ranges(True, hybrid=mode)
                ctx.prof_reset()
                ctx.prof_enable(True)
                got[name, mode] = corpus.store.hybrid_search(st, q, limit, sim, refs, n_tok, float(n_docs), k, thr, allow=bm if filt else None, rescale_e5=e5)
                ctx.prof_enable(False)
                if mode:
                    served += ctx.prof_get("bm25_range_score")[1] > 0 and ctx.prof_get("bm25_accumulate")[1] == 0
                else:
                    assert ctx.prof_get("bm25_accumulate")[1] > 0
        ids0, sc0, cnt0 = got["plain", False]
        for key, (ids, sc, cnt) in got.items():
            assert cnt == cnt0 and ids.tolist() == ids0.tolist() and np.array_equal(bits(sc), bits(sc0)), (case, key)
        # ... and the oracle's combine over the oracle's epilogue of the library's own vector hits (the scan itself is
        # checked against the oracle in test_vector_gpu.py)
        h_ids, h_dist, h_n = stores["plain"].storage_search(q, limit, bm if filt else None)
        vec = orc.embedding_epilogue(h_ids[0][: h_n[0]], h_dist[0][: h_n[0]], e5, sim)
        fd, fs = orc.search_full_text(corpus.entries(refs, allow_mask if filt else None), n_tok, float(n_docs), 1.2, thr)
        od, os_ = orc.normalize_and_combine(list(vec), [float(v) for v in vec.values()], fd, fs)
        td, ts = orc.top_n(od, os_, k)
        assert cnt0 == len(od) and ids0.tolist() == td.tolist(), case
        assert np.array_equal(bits(sc0), bits(ts)), case
    assert served >= 14  # the range scorer answered (nearly) every eligible call without the per-record scorer
    ctx.set_bm25_ranges(True)

    # ties at the cut -> the candidates cannot prove the answer -> the per-record scorer answers, same result
    docs = np.arange(3000, dtype=np.int64)
    flat = Corpus(ctx, 3000, [(0, docs)], [7.0], seed=1)
    flat.store.build(flat.doc_ids, flat.avg, [ft.PostingList(field=0, docs=flat.doc_ids, tf=np.ones(3000), field_len=np.full(3000, 7))])
    fv = oa.EmbeddingFieldStorage(ctx, dimensions=dim)
    fv.insert_rows(flat.doc_ids, rows[:3000])
    res = {}
    for mode in (True, False):
        ctx.set_bm25_ranges(True, hybrid=mode)
        res[mode] = flat.store.hybrid_search(fv, rows[17], 5, 0.0, [(0, 0, 1.0)], 1, 3000.0, 10)
    assert res[True][2] == res[False][2] == 3000 and res[True][0].tolist() == res[False][0].tolist()
    assert np.array_equal(bits(res[True][1]), bits(res[False][1]))
    # OMC -> per-record scorer; no full-text side -> the vector map alone
    ctx.set_bm25_ranges(True)
    flat.store.set_omc({11: 4.0})
    a = flat.store.hybrid_search(fv, rows[17], 5, 0.0, [(0, 0, 1.0)], 1, 3000.0, 10, apply_omc=True)
    assert 11 in a[0].tolist()
    flat.store.set_omc({})
    e = flat.store.hybrid_search(fv, rows[17], 5, 0.0, [], 1, 3000.0, 10)
    assert e[2] == 5 and e[0][0] == 17
    # a vector hit that is not a document of the index: the per-record scorer's error, not a wrong answer
    fv.insert_rows(np.array([999_999], dtype=np.uint64), rows[17:18] * F(1.0))
    with pytest.raises(oa.OramaError):
        flat.store.hybrid_search(fv, rows[17], 5, 0.0, [(0, 0, 1.0)], 1, 3000.0, 10)
    fv.close()
    flat.store.close()
    for st in stores.values():
        st.close()
    corpus.store.close()


def test_large_lists_and_multi_chunk_batches(ctx):
    """A 600 000-document store with lists of up to 200 000 postings (hundreds of ranges per query, key lists of several
    reduction chunks): K3r == K3 == the oracle, bit for bit — plain, thresholds, filter, OMC, hybrid, k from 1 to 300 —
    and a batch of 70 queries (three sets of launches, two in flight at a time on two scratch sets) == the same queries
    one by one; 300 000 documents with one and the same score keep the (score desc, DocumentId asc) cut."""
    rng = np.random.default_rng(91)
    n_docs = 600_000
    sizes = [200_000, 120_000, 90_000, 60_000, 40_000, 30_000, 20_000, 9_000, 3_000, 500]
    lists = [(i % 2, np.sort(rng.choice(n_docs, size=sz, replace=False))) for i, sz in enumerate(sizes)]
    corpus = Corpus(ctx, n_docs, lists, [70.0, 11.0], seed=92)
    allow_mask = rng.random(n_docs) < 0.5
    bm = oa.AllowBitmap(n_docs, np.nonzero(allow_mask)[0].astype(np.uint64))
    omc = {int(d): float(F(m)) for d, m in zip(rng.choice(n_docs, size=50, replace=False), rng.uniform(0.5, 40.0, size=50))}

    def run(refs, n_tok, k, thr=None, filt=False, use_omc=False):
        return corpus.store.search(refs, n_tok, float(n_docs), k, thr, allow=bm if filt else None, apply_omc=use_omc)

    cases = [
        ([(0, 0, 1.0), (1, 1, 1.0), (2, 3, 2.0)], 3, 100, None, False, False),
        ([(0, 0, 1.0), (1, 2, 1.0), (2, 4, 1.0), (3, 7, 1.5)], 4, 10, 2, False, False),
        ([(0, 1, 1.0), (0, 2, 1.0), (1, 0, 1.0)], 2, 256, None, False, False),  # two lists of one token: df on the device
        ([(0, 0, 1.0), (1, 1, 1.0)], 2, 1, None, True, False),
        ([(0, 0, 1.0), (1, 5, 1.0), (2, 9, 1.0)], 3, 50, None, False, True),
        ([(0, 3, 1.0), (1, 6, 1.0)], 2, 300, None, False, False),
        ([(0, 8, 1.0), (1, 9, 1.0)], 2, 20, None, False, False),
    ]
    for ci, (refs, n_tok, k, thr, filt, use_omc) in enumerate(cases):
        corpus.store.set_omc(omc if use_omc else {})
        od, os_, ocount = corpus.oracle(refs, n_tok, k, thr, allow_mask if filt else None, omc if use_omc else None)
        got = {}
        for name, on in (("k3r", True), ("k3", False)):
            ctx.set_bm25_ranges(on)
            got[name] = run(refs, n_tok, k, thr, filt, use_omc)
        for name, (ids, sc, count) in got.items():
            assert count == ocount and ids.tolist() == od.tolist() and np.array_equal(bits(sc), bits(os_)), (ci, name)
    corpus.store.set_omc({})
    # hybrid (k + n_vec + 1 candidates)
    ctx.set_bm25_ranges(True)
    refs = [(0, 0, 1.0), (1, 1, 1.0), (2, 2, 1.0)]
    fd, fs = orc.search_full_text(corpus.entries(refs), 3, float(n_docs), 1.2, None)
    vec = {int(d): float(F(s)) for d, s in zip(np.concatenate([rng.choice(fd, 40, replace=False), rng.choice(n_docs, 30, replace=False).astype(np.uint64)]),
                                               rng.uniform(0.0, 1.0, size=70))}
    od, os_ = orc.normalize_and_combine(list(vec), list(vec.values()), fd, fs)
    td, ts = orc.top_n(od, os_, 60)
    ids, sc, count = corpus.store.search(refs, 3, float(n_docs), 60, vector=vec, apply_omc=False)
    assert count == len(od) and ids.tolist() == td.tolist() and np.array_equal(bits(sc), bits(ts))
    # a batch over three chunks, eligible and small queries interleaved
    queries = []
    for i in range(70):
        if i % 3 == 2:
            queries.append(([(0, 8, 1.0), (1, 9, 1.0)], 2, None, 5 + i))
        else:
            queries.append(([(0, int(i % 4), 1.0), (1, 4 + int(i % 3), 1.0)], 2, 1 if i % 5 == 0 else None, 1 + (i * 7) % 200))
    ctx.set_bm25_ranges(True)
    a = corpus.store.search_batch(queries, float(n_docs), 256)
    b = [corpus.store.search(refs, n_tok, float(n_docs), k, thr) for refs, n_tok, thr, k in queries]
    for i, ((ia, sa, ca), (ib, sb, cb)) in enumerate(zip(a, b)):
        assert ca == cb and ia.tolist() == ib.tolist() and np.array_equal(bits(sa), bits(sb)), i
        if i < 6:
            refs, n_tok, thr, k = queries[i]
            od, os_, ocount = corpus.oracle(refs, n_tok, k, thr)
            assert ca == ocount and ia.tolist() == od.tolist() and np.array_equal(bits(sa), bits(os_)), i
    corpus.store.close()
    # every document scores the same
    n_flat = 300_000
    docs = np.arange(n_flat, dtype=np.int64)
    flat = Corpus(ctx, n_flat, [(0, docs)], [7.0], seed=1)
    flat.store.build(flat.doc_ids, flat.avg, [ft.PostingList(field=0, docs=flat.doc_ids, tf=np.ones(n_flat), field_len=np.full(n_flat, 7))])
    ids, sc, count = flat.store.search([(0, 0, 1.0)], 1, float(n_flat), 25)
    assert count == n_flat and ids.tolist() == list(range(25)) and len(set(bits(sc).tolist())) == 1
    flat.store.close()


def test_compact_key_lists_equal_one_slot_per_posting(ctx):
    """Round 5: a plain top-k batch APPENDS only the keys at or above a floor (bm25_ranges.hip, COMPACT) instead of writing one
    slot per posting.  A floor is a score at least `top_k` documents are known to reach, so it cannot cut the answer: the same
    ids, score bits and counts as round 4's lists (orama_ctx_set_bm25_ranges(ctx, 3)) and as the oracle — for k from 1 to past
    the 256 the floor supports, for lists full of EQUAL scores (every key ties at the floor), for count-only queries, under a
    filter and a threshold, single calls and batches."""
    rng = np.random.default_rng(55)
    n_docs = 600_000
    lists = random_lists(rng, n_docs, 10, 2, 20_000, 90_000)
    corpus = Corpus(ctx, n_docs, lists, [64.0, 9.5], seed=56)
    # two lists whose postings all carry the same tf and length: thousands of equal scores, ties decided by document id
    same = np.sort(rng.choice(n_docs, size=50_000, replace=False))
    flat = Corpus(ctx, n_docs, [(0, same), (0, same[::2])], [10.0], seed=57)
    flat.store.close()  # (the same lists again with constant tf and length)
    flat.lists = [(f, loc, np.full(len(loc), 2), np.full(len(loc), 10)) for f, loc, _, _ in flat.lists]
    flat.store = ft.PostingsStore(ctx)
    flat.store.build(flat.doc_ids, flat.avg, [ft.PostingList(field=f, docs=flat.doc_ids[loc], tf=tf, field_len=ln)
                                              for f, loc, tf, ln in flat.lists])
    allow_mask = rng.random(n_docs) < 0.5
    bm = oa.AllowBitmap(n_docs, np.arange(n_docs, dtype=np.uint64)[allow_mask])
    cases = []
    for k in (1, 7, 100, 256, 257, 300, 1000):
        refs = [(t, int(l), 1.0) for t, l in enumerate(rng.choice(10, size=6, replace=False))]
        cases.append((corpus, refs, 6, k, None, None, None))
    cases.append((corpus, [(0, 1, 1.0), (1, 2, 2.0), (2, 3, 1.0)], 3, 50, 2, None, None))       # threshold
    cases.append((corpus, [(0, 4, 1.0), (1, 5, 1.0)], 2, 100, None, bm, allow_mask))           # filter
    cases.append((flat, [(0, 0, 1.0)], 1, 100, None, None, None))                                # all scores equal
    cases.append((flat, [(0, 0, 1.0), (1, 1, 1.0)], 2, 100, None, None, None))                   # two score levels, many ties
    cases.append((corpus, [(0, 0, 1.0), (1, 9, 1.0)], 2, 0, None, None, None))                   # count only
    answers = {}
    for compact in ("always", False):  # ("always": single calls too — by default only batches of 8 queries and more)
        ctx.set_bm25_ranges(True, compact_keys=compact)
        got = []
        for c, refs, nt, k, thr, allow, mask in cases:
            ids, sc, count = c.store.search(refs, nt, float(n_docs), k, thr, allow=allow)
            if k:
                od, os_, ocount = c.oracle(refs, nt, k, thr, mask)
                assert count == ocount and ids.tolist() == od.tolist() and np.array_equal(bits(sc), bits(os_)), (compact, refs, k)
            got.append((ids, sc, count))
        # the same queries of the big corpus as ONE batch (queries of different k share a set of launches: k = the largest)
        batch = [(refs, nt, thr) for c, refs, nt, k, thr, allow, mask in cases if c is corpus and allow is None]
        got.append(corpus.store.search_batch(batch, float(n_docs), 100))
        answers[compact] = got
    ctx.set_bm25_ranges(True)  # (the scorer setter leaves the key-list form alone since round 6: ADVICE r05)
    ctx.set_bm25_ranges(True, compact_keys=True)  # ... back to the default explicitly
    for a, b_ in zip(answers["always"][:-1], answers[False][:-1]):
        assert a[2] == b_[2] and a[0].tolist() == b_[0].tolist() and np.array_equal(bits(a[1]), bits(b_[1]))
    for a, b_ in zip(answers["always"][-1], answers[False][-1]):
        assert a[2] == b_[2] and a[0].tolist() == b_[0].tolist() and np.array_equal(bits(a[1]), bits(b_[1]))
    corpus.store.close()
    flat.store.close()


def test_the_range_width_that_held_is_remembered(ctx):
    """lists that overlap heavily (the same term in two fields) overflow the scoring launch's cell tables at the
    default range width; the query is rerun with narrower ranges — and the width that held is remembered for the lists, so
    the next query over them is scored once.  Same answers before and after, equal to the oracle's."""
    rng = np.random.default_rng(77)
    n_docs = 200_000
    base = np.sort(rng.choice(n_docs, size=120_000, replace=False))
    lists = [(0, base), (1, base), (0, np.sort(rng.choice(n_docs, size=30_000, replace=False)))]
    corpus = Corpus(ctx, n_docs, lists, [40.0, 6.0], seed=78)
    refs = [(0, 0, 1.0), (0, 1, 2.0), (1, 2, 1.0)]  # token 0 in both fields: every document of `base` has two postings
    od, os_, ocount = corpus.oracle(refs, 2, 100)
    for _ in range(3):  # first call: overflow + rerun; later calls start at the remembered width
        ids, sc, count = corpus.store.search(refs, 2, float(n_docs), 100)
        assert count == ocount and ids.tolist() == od.tolist() and np.array_equal(bits(sc), bits(os_))
    res = corpus.store.search_batch([(refs, 2, None)] * 40, float(n_docs), 100)
    for ids, sc, count in res:
        assert count == ocount and ids.tolist() == od.tolist() and np.array_equal(bits(sc), bits(os_))
    corpus.store.close()


def test_postings_per_range_never_change_an_answer(ctx):
    """Option `k3r_target` (postings per document range; 0 = the default, 7/8 of what a scoring workgroup holds): from ranges of a
    few dozen postings to ranges AT the workgroup's capacity (where many overflow and their queries rerun with narrower ranges),
    single calls and batches return the oracle's ids, score bits and counts.  The round-6 experiment `k3r_fast` is a comparison
    unit: the product library refuses the option instead of ignoring it."""
    rng = np.random.default_rng(91)
    n_docs = 300_000
    lists = random_lists(rng, n_docs, 10, 2, 500, 60_000)
    corpus = Corpus(ctx, n_docs, lists, [50.0, 8.0], seed=92)
    queries = []
    for _ in range(12):
        nt = int(rng.integers(1, 7))
        queries.append(([(t, int(l), float(F(rng.choice([1.0, 2.0])))) for t, l in enumerate(rng.choice(10, size=nt, replace=False))], nt, None))
    expect = [corpus.oracle(refs, nt, 100) for refs, nt, _ in queries]
    try:
        for target in (0, 16, 300, 1792, 2048, 4096):  # (4096: above what a 256-thread workgroup holds -> the default)
            ctx.set_option("k3r_target", target)
            got = corpus.store.search_batch(queries, float(n_docs), 100)
            got += [corpus.store.search(refs, nt, float(n_docs), 100) for refs, nt, _ in queries[:4]]
            for (ids, sc, count), (od, os_, ocount) in zip(got, expect + expect[:4]):
                assert count == ocount and ids.tolist() == od.tolist() and np.array_equal(bits(sc), bits(os_)), target
    finally:
        ctx.set_option("k3r_target", 0)
    from oramacore_amd import _build

    if not _build.comparison_build():
        with pytest.raises(oa.OramaError):
            ctx.set_option("k3r_fast", 1)
    corpus.store.close()


_FAST_BODY_SCRIPT = r"""
import sys
root = sys.argv[1]
sys.path.insert(0, root); sys.path.insert(0, root + "/tests")
import numpy as np
import oramacore_amd as oa
from test_bm25_ranges_gpu import Corpus, random_lists, bits, F

rng = np.random.default_rng(123)
n_docs = 600_000
lists = random_lists(rng, n_docs, 12, 2, 2_000, 120_000)
allow_mask = rng.random(n_docs) < 0.7
queries = []
for i in range(96):
    nt = int(rng.integers(2, 9))
    refs = [(t, int(l), float(F(rng.choice([1.0, 1.5])))) for t, l in enumerate(rng.choice(12, size=nt, replace=False))]
    if i % 8 == 0:  # the same token in two lists (two fields): multi-posting documents whose postings share a token
        refs.append((refs[-1][0], int((refs[-1][1] + 1) % 12), 1.0))
    queries.append((refs, nt, 2 if i % 11 == 0 else None))  # (threshold: at least 2 of the query's tokens)
answers = {}
for fast in (0, 1):
    ctx = oa.Context(0)
    ctx.set_option("k3r_fast", fast)
    ctx.set_option("bm25_dense_acc", fast)  # (bitmaps of the longest lists: what the fast body's background lists read)
    corpus = Corpus(ctx, n_docs, lists, [50.0, 8.0], seed=124)
    bm = oa.AllowBitmap(n_docs, np.arange(n_docs, dtype=np.uint64)[allow_mask])
    got = {}
    for k in (1, 100, 300):
        got[("plain", k)] = corpus.store.search_batch(queries, float(n_docs), k)
    got[("filtered", 100)] = corpus.store.search_batch(queries[:48], float(n_docs), 100, allow=bm)
    if fast:
        for (refs, nt, thr), (ids, sc, count) in zip(queries[:6], got[("plain", 100)][:6]):
            od, os_, ocount = corpus.oracle(refs, nt, 100, thr)
            assert count == ocount and ids.tolist() == od.tolist() and np.array_equal(bits(sc), bits(os_))
    answers[fast] = got
    corpus.store.close(); ctx.close()
for key in answers[0]:
    for a, b in zip(answers[0][key], answers[1][key]):
        assert a[2] == b[2] and a[0].tolist() == b[0].tolist() and np.array_equal(bits(a[1]), bits(b[1])), key
print("identical", sum(len(v) for v in answers[0].values()))
"""


def test_comparison_scoring_body_equals_the_product_body(tmp_path):
    """bm25_ranges_fast.hip (round 6: singletons without ranks, lists under the published floor left unscored) is a comparison
    unit — not faster, kept with its record (profiles/r06_k3r_fast_body.md).  Where liborama_hip_cmp.so is present it must return
    the product body's answers bit for bit: batches of 96 queries (floors get published: hundreds of ranges per query), k = 1 /
    100 / 300, thresholds, a token with two lists, a filter; the first queries against the oracle as well."""
    import os, subprocess, sys
    from pathlib import Path

    from oramacore_amd import _build

    if not _build.LIB_CMP.exists():
        pytest.skip("liborama_hip_cmp.so is not built (ORAMA_COMPARISON_KERNELS=1 python -c 'import __graft_entry__ as g; g.build()')")
    root = str(Path(__file__).resolve().parent.parent)
    script = tmp_path / "fast_body.py"
    script.write_text(_FAST_BODY_SCRIPT)
    r = subprocess.run([sys.executable, str(script), root], env=dict(os.environ, ORAMA_COMPARISON_KERNELS="1"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    assert r.stdout.strip().splitlines()[-1].startswith("identical"), r.stdout[-500:]


def test_hybrid_tail_on_the_device_equals_the_host_tail():
    """Round 5: orama_hybrid_search finishes ON THE DEVICE (hybrid_tail.hip: a2 epilogue, per-document scoring of the hits,
    normalize_and_combine, K4 — one read-back, one host wake-up).  Same corpus on two contexts, one of them with option
    "hybrid_device_tail" = 1, the other with the default (round 3's host tail): identical ids, score bits and counts — plain and shadow stores, E5
    rescale, cut-off, filter, several rows per document, thresholds, k / limit from 1 to 300 (beyond 512 hits the device form
    steps aside by itself)."""
    import os

    import util

    rng = np.random.default_rng(177)
    n_docs, dim = 20_000, 64
    lists = random_lists(rng, n_docs, 12, 2, 200, 5000)
    doc_ids = np.arange(n_docs, dtype=np.uint64) * 2 + 9
    rows = util.gaussian_rows(n_docs + 3000, dim, seed=179)
    row_doc = np.concatenate([doc_ids, doc_ids[rng.choice(n_docs, size=3000, replace=False)]])
    allow_mask = rng.random(n_docs) < 0.6
    cases = []
    for case in range(8):
        n_tok = int(rng.integers(1, 5))
        refs = [(t, int(l), float(F(rng.choice([1.0, 1.5])))) for t in range(n_tok) for l in rng.choice(12, size=int(rng.integers(1, 3)), replace=False)]
        q = (rows[int(rng.integers(len(rows)))] + F(0.4) * rows[int(rng.integers(len(rows)))]).astype(F)
        k, limit = [(10, 10), (1, 300), (100, 100), (300, 1), (25, 40), (100, 7), (5, 64), (50, 200)][case]
        sim, e5, filt = [(0.0, False, False), (0.0, False, True), (0.5, False, False), (0.0, True, False)][case % 4]
        cases.append((refs, n_tok, q, k, limit, sim, e5, filt, 1 if case == 5 else None))
    answers = {}
    for form in ("device", "host"):
        c = oa.Context(0)
        try:
            c.set_option("hybrid_device_tail", 1 if form == "device" else 0)
        except oa.OramaError as e:  # the product library does not carry hybrid_tail.hip (a comparison unit since round 6)
            assert form == "device" and e.status == oa._native.ORAMA_ERR_UNSUPPORTED, e
            pytest.skip("the device hybrid tail is built into liborama_hip_cmp.so only (ORAMA_COMPARISON_KERNELS=1)")
        corpus = Corpus(c, n_docs, lists, [80.0, 12.0], doc_ids=doc_ids, seed=178)
        bm = oa.AllowBitmap(int(doc_ids.max()) + 1, doc_ids[allow_mask])
        stores = {"plain": oa.EmbeddingFieldStorage(c, dimensions=dim), "shadow": oa.EmbeddingFieldStorage(c, dimensions=dim, dtype=oa._native.DTYPE_F32_SHADOW16)}
        c.set_two_stage(True, always=True)  # (a 20 K-row store would take the plain scan otherwise)
        got = []
        for st in stores.values():
            st.insert_rows(row_doc, rows)
            for refs, n_tok, q, k, limit, sim, e5, filt, thr in cases:
                got.append(corpus.store.hybrid_search(st, q, limit, sim, refs, n_tok, float(n_docs), k, thr, allow=bm if filt else None, rescale_e5=e5))
        answers[form] = got
        for st in stores.values():
            st.close()
        corpus.store.close()
        c.close()
    for i, (a, b_) in enumerate(zip(answers["device"], answers["host"])):
        assert a[2] == b_[2] and a[0].tolist() == b_[0].tolist() and np.array_equal(bits(a[1]), bits(b_[1])), i
