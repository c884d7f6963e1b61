"""Live updates of the resident postings (orama_post_append): a store built from the first part
of the documents and then appended to must answer exactly like a store built from all documents at once."""
import numpy as np
import pytest

import oramacore_amd as oa
from oramacore_amd import fulltext as ft

pytestmark = pytest.mark.gpu


def make_corpus(n, n_terms, seed, id_mul=1, id_add=0):
    rng = np.random.default_rng(seed)
    doc_ids = np.arange(n, dtype=np.uint64) * np.uint64(id_mul) + np.uint64(id_add)
    lens = rng.integers(5, 300, size=n).astype(np.uint32)
    terms = []
    for _ in range(n_terms):
        pos = np.sort(rng.choice(n, size=int(rng.integers(n // 20, n // 2)), replace=False))
        terms.append((pos, rng.integers(1, 6, size=len(pos)).astype(np.uint32)))
    return doc_ids, lens, terms


def lists_for(doc_ids, lens, terms, lo, hi):
    out = []
    for pos, tf in terms:
        keep = (pos >= lo) & (pos < hi)
        out.append(ft.PostingList(field=0, docs=doc_ids[pos[keep]], tf=tf[keep], field_len=lens[pos[keep]]))
    return out


@pytest.mark.parametrize("id_mul,id_add", [(1, 0), (3, 7)], ids=["dense_ids", "sparse_ids"])
def test_append_equals_rebuild(ctx, id_mul, id_add):
    n, T = 6000, 5
    doc_ids, lens, terms = make_corpus(n, T, seed=21, id_mul=id_mul, id_add=id_add)
    avg_all = float(lens.mean())
    whole = ft.PostingsStore(ctx)
    whole.build(doc_ids, [avg_all], lists_for(doc_ids, lens, terms, 0, n))
    cuts = [0, 3500, 3501, 5200, n]
    live = ft.PostingsStore(ctx)
    live.build(doc_ids[:cuts[1]], [float(lens[:cuts[1]].mean())], lists_for(doc_ids, lens, terms, 0, cuts[1]))
    live.set_omc({int(doc_ids[10]): 3.0})
    refs_live = [[(t, t, 1.0)] for t in range(T)]
    for a, b in zip(cuts[1:-1], cuts[2:]):
        first = live.append(doc_ids[a:b], [float(lens[:b].mean())], lists_for(doc_ids, lens, terms, a, b))
        assert first == live.info()["n_lists"] - T
        for t in range(T):
            refs_live[t].append((t, first + t, 1.0))
    assert live.info()["n_docs"] == n and live.info()["n_postings"] == whole.info()["n_postings"]
    whole.set_omc({int(doc_ids[10]): 3.0})
    refs_whole = [(t, t, 1.0) for t in range(T)]
    flat = [r for t in range(T) for r in refs_live[t]]
    bm = oa.AllowBitmap.from_mask((np.arange(int(doc_ids.max()) + 1) % 7) != 2)
    for allow in (None, bm):
        for thr in (None, 3):
            a = whole.search(refs_whole, T, float(n), 200, thr, allow=allow)
            b = live.search(flat, T, float(n), 200, thr, allow=allow)
            assert a[2] == b[2] and a[0].tolist() == b[0].tolist()
            assert np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32))
    vec = {int(doc_ids[5]): 0.8, int(doc_ids[n - 1]): 0.3}  # one committed doc, one appended doc
    a = whole.search(refs_whole, T, float(n), 100, vector=vec)
    b = live.search(flat, T, float(n), 100, vector=vec)
    assert a[2] == b[2] and a[0].tolist() == b[0].tolist() and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32))
    whole.close()
    live.close()


def test_append_validation(ctx):
    doc_ids, lens, terms = make_corpus(100, 2, seed=3)
    st = ft.PostingsStore(ctx)
    with pytest.raises(oa.OramaError):  # nothing built yet
        st.append(doc_ids[:10], [10.0], [])
    st.build(doc_ids[:50], [10.0], lists_for(doc_ids, lens, terms, 0, 50))
    with pytest.raises(oa.OramaError):  # ids must continue upwards
        st.append(doc_ids[40:60], [10.0], [])
    with pytest.raises(oa.OramaError):  # posting of an unknown document
        st.append(doc_ids[50:60], [10.0], [ft.PostingList(field=0, docs=np.array([99], dtype=np.uint64),
                                                          tf=np.array([1]), field_len=np.array([5]))])
    assert st.info()["n_docs"] == 50
    st.append(doc_ids[50:60], [10.0], [])
    assert st.info()["n_docs"] == 60
    st.close()


def test_searches_concurrent_with_appends(ctx):
    """update_data(&self) runs while searches hold the read lock (index/mod.rs:1436): searches issued while another
    thread appends documents must each see a consistent store — the answer before or after an append, never a mix."""
    import threading

    n, T = 4000, 3
    doc_ids, lens, terms = make_corpus(n, T, seed=33)
    avg = float(lens.mean())
    cuts = list(range(2000, n + 1, 250))
    live = ft.PostingsStore(ctx)
    live.build(doc_ids[:cuts[0]], [avg], lists_for(doc_ids, lens, terms, 0, cuts[0]))
    # expected answers for every prefix length, from separately built stores
    expected = {}
    for hi in cuts:
        ref = ft.PostingsStore(ctx)
        ref.build(doc_ids[:hi], [avg], lists_for(doc_ids, lens, terms, 0, hi))
        ids, sc, count = ref.search([(t, t, 1.0) for t in range(T)], T, float(n), 30)
        expected[count] = (ids.tolist(), sc.view(np.uint32).tolist())
        ref.close()
    refs_now = [[(t, t, 1.0)] for t in range(T)]
    lock = threading.Lock()
    stop = threading.Event()
    errors = []

    def searcher():
        while not stop.is_set():
            with lock:
                flat = [r for t in range(T) for r in refs_now[t]]
            try:
                ids, sc, count = live.search(flat, T, float(n), 30)
            except Exception as e:  # noqa: BLE001
                errors.append(e)
                return
            # refs may lag the store by one append (then the newest delta lists are simply not referenced yet):
            # the count identifies which prefix was answered, and that answer must be exact
            if count not in expected or expected[count] != (ids.tolist(), sc.view(np.uint32).tolist()):
                errors.append(AssertionError(f"inconsistent answer for count {count}"))
                return

    threads = [threading.Thread(target=searcher) for _ in range(4)]
    for th in threads:
        th.start()
    for a, b in zip(cuts[:-1], cuts[1:]):
        first = live.append(doc_ids[a:b], [avg], lists_for(doc_ids, lens, terms, a, b))
        with lock:
            for t in range(T):
                refs_now[t].append((t, first + t, 1.0))
    stop.set()
    for th in threads:
        th.join()
    assert not errors, errors[:2]
    ids, sc, count = live.search([r for t in range(T) for r in refs_now[t]], T, float(n), 30)
    assert expected[count] == (ids.tolist(), sc.view(np.uint32).tolist()) and count == max(expected)
    live.close()
