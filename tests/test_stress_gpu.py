"""Everything at once for a few seconds: two-stage searches, plain searches, request batchers (vector + full-text),
hybrid, BM25 batches, live inserts, deletes and compactions of a shadow store — looking for deadlocks (lock orders of
the store pair, the scratch pool's multi-set waiters) and for answers that break invariants (sorted, complete, only
published documents).  Exact parity is the business of the other tests."""
import threading
import time

import numpy as np
import pytest

import oramacore_amd as oa
from oramacore_amd import fulltext as ft

pytestmark = pytest.mark.gpu


def test_mixed_load_does_not_deadlock_or_corrupt():
    ctx = oa.Context(0)
    ctx.set_two_stage(True, always=True)
    rng = np.random.default_rng(77)
    dim, n0 = 128, 60_000
    rows = rng.standard_normal((n0, dim)).astype(np.float32)
    ids = np.arange(n0, dtype=np.uint64)
    shadow = oa.EmbeddingFieldStorage(ctx, dimensions=dim, dtype=oa.DTYPE_F32_SHADOW16)
    shadow.insert_rows(ids, rows)
    lists = []
    for _ in range(16):
        local = np.sort(rng.choice(n0, size=int(rng.integers(500, 6000)), replace=False))
        lists.append(ft.PostingList(field=0, docs=ids[local], tf=rng.integers(1, 5, size=len(local)),
                                    field_len=rng.integers(5, 200, size=len(local))))
    post = ft.PostingsStore(ctx)
    post.build(ids, [60.0], lists)
    first_docs = oa.AllowBitmap(n0, ids).to_device(ctx)  # the documents the postings store knows (rows are added during the run)
    vb = oa.SearchBatcher(shadow, max_batch=64)
    pb = ft.PostSearchBatcher(post, max_batch=64)
    stop = threading.Event()
    errors = []
    counts = {"vec": 0, "ft": 0, "hyb": 0, "batch": 0, "ins": 0, "del": 0, "compact": 0}
    lock = threading.Lock()

    def guard(fn):
        def run():
            try:
                while not stop.is_set():
                    fn()
            except Exception as e:  # noqa: BLE001
                errors.append(e)
                stop.set()
        return run

    def bump(k):
        with lock:
            counts[k] += 1

    def vec_worker():
        r = np.random.default_rng(threading.get_ident() % 2**32)
        q = r.standard_normal((int(r.integers(1, 20)), dim)).astype(np.float32)
        if r.random() < 0.5:
            i, d, c = shadow.storage_search(q, 10)
            assert np.all(c == 10) and np.all(np.diff(d, axis=1) >= 0)
        else:
            i, d = vb.search(q[0], 10)
            assert len(i) == 10 and np.all(np.diff(d) >= 0)
        bump("vec")

    def ft_worker():
        r = np.random.default_rng(threading.get_ident() % 2**32)
        refs = [(t, int(r.integers(0, 16)), 1.0) for t in range(int(r.integers(1, 5)))]
        if r.random() < 0.5:
            i, s, c = pb.search(refs, len(refs), float(n0), 10)
        else:
            i, s, c = post.search(refs, len(refs), float(n0), 10)
        assert c >= len(i) and np.all(np.diff(s) <= 0)
        bump("ft")

    def hybrid_worker():
        r = np.random.default_rng(threading.get_ident() % 2**32)
        refs = [(t, int(r.integers(0, 16)), 1.0) for t in range(3)]
        q = r.standard_normal(dim).astype(np.float32)
        if r.random() < 0.5:
            i, s, c = post.hybrid_search(shadow, q, 20, 0.0, refs, 3, float(n0), 10, allow=first_docs)
        else:
            vi, vd = vb.search(q, 20)
            keep = vi < n0  # documents inserted during the run are not in the postings store
            i, s, c = post.search(refs, 3, float(n0), 10, vector=dict(zip(vi[keep].tolist(), (1.0 - vd[keep]).tolist())), apply_omc=False)
        assert len(i) <= 10 and np.all(np.diff(s) <= 0)
        bump("hyb")

    def batch_worker():
        r = np.random.default_rng(threading.get_ident() % 2**32)
        qs = [([(t, int(r.integers(0, 16)), 1.0) for t in range(2)], 2, None) for _ in range(40)]
        res = post.search_batch(qs, float(n0), 10)
        assert len(res) == 40
        bump("batch")

    next_id = [n0]

    def mutate_worker():
        r = np.random.default_rng(5)
        more = r.standard_normal((3000, dim)).astype(np.float32)
        mids = np.arange(next_id[0], next_id[0] + 3000, dtype=np.uint64)
        shadow.insert_rows(mids, more)
        next_id[0] += 3000
        bump("ins")
        for d in r.choice(n0, size=20, replace=False):
            shadow.delete(int(d))
        bump("del")
        if counts["ins"] % 3 == 0:
            shadow.compact(counts["ins"])
            bump("compact")
        time.sleep(0.01)

    workers = [vec_worker] * 4 + [ft_worker] * 3 + [hybrid_worker] * 3 + [batch_worker] + [mutate_worker]
    ths = [threading.Thread(target=guard(w)) for w in workers]
    for t in ths:
        t.start()
    time.sleep(6.0)
    stop.set()
    for t in ths:
        t.join(timeout=60)
    assert not any(t.is_alive() for t in ths), "a worker is stuck (deadlock)"
    assert not errors, errors
    assert all(v > 0 for v in counts.values()), counts
    info = shadow.info()
    assert info["num_rows"] == info["num_embeddings"] + info["pending_ops"]
    vb.close()
    pb.close()
    post.close()
    shadow.close()
    ctx.close()
