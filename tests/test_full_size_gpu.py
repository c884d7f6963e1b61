"""Parity at BASELINE.json's FULL sizes through size-independent properties (the oracle cannot scan 30 GB in
seconds): NS 10 M x 768 fp32 (Q=1), C3 10 M x 768 fp16 (Q=64), C4-scale BM25 (10 M docs, 12 tokens).

Properties: sortedness; planted exact matches surface on top; reported distances equal the oracle's on the rows
read back (1e-4); no row of a random sample beats the k-th distance; idempotence; shard/merge identity (the
multi-GPU reduction) and batch == solo.  BM25 at full size is checked against the oracle itself, bit-exact (the
oracle handles 600 K postings in about a second)."""
import numpy as np
import pytest

import oramacore_amd as oa
import util
from oracle import oracle as orc
from oramacore_amd import _native as N
from oramacore_amd import fulltext as ft

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _check_store(st, n, d, k, q, quantised: bool):
    plant = np.stack([q * np.float32(s) for s in (0.5, 1.0, 3.0)])
    st.insert_rows(np.arange(n, n + 3, dtype=np.uint64), plant)
    ids, dist, cnt = st.storage_search(q, k)
    assert cnt[0] == k and np.all(np.diff(dist[0]) >= 0)
    zero_tol = 2e-3 if quantised else 1e-6
    assert sorted(ids[0, :3].tolist()) == [n, n + 1, n + 2] and np.all(np.abs(dist[0, :3]) < zero_tol)
    ids2, dist2, _ = st.storage_search(q, k)                                    # idempotence
    assert np.array_equal(ids, ids2) and np.array_equal(dist, dist2)
    rows, docs = st.get_rows(ids[0])                                            # the rows as stored (f16 → f32)
    assert np.array_equal(docs, ids[0])
    od = orc.distances(rows, q)
    assert np.max(np.abs(od - dist[0])) <= TOL
    rng = np.random.default_rng(3)
    sample = rng.choice(n, size=50_000, replace=False).astype(np.uint64)
    srows, _ = st.get_rows(sample)
    sd = orc.distances(srows, q, threads=8)
    inside = set(ids[0].tolist())
    worse = sd >= dist[0, -1] - 2 * TOL
    assert all(w or int(r) in inside for w, r in zip(worse.tolist(), sample.tolist()))
    return ids, dist


def test_ns_full_size_fp32(ctx):
    n, d, k = 10_000_000, 768, 100
    st = oa.EmbeddingFieldStorage(ctx, dimensions=d, reserve_rows=n + 16)
    st.fill_synthetic(n, seed=0xC0FFEE)
    q = util.gaussian_rows(1, d, seed=0xBEEF)[0]
    ids, dist = _check_store(st, n, d, k, q, quantised=False)
    # shard/merge identity over 8 interleaved shards == the 8-GPU reduction
    parts_i, parts_d = [], []
    for g in range(8):
        bm = oa.AllowBitmap.from_mask(np.arange(n + 3) % 8 == g).to_device(ctx)
        i, v, c = st.storage_search(q, k, bm)
        parts_i.append(i[0, :c[0]])
        parts_d.append(v[0, :c[0]])
        bm.close()
    md, ms = orc.top_n(np.concatenate(parts_i), -np.concatenate(parts_d), k)
    assert np.array_equal(md, ids[0]) and np.array_equal(-ms, dist[0])
    # a batch of 5 (K1b) equals the solo answers
    qs = util.gaussian_rows(5, d, seed=77)
    bi, bd, bc = st.storage_search(qs, k)
    for j in (0, 4):
        si, sd, sc = st.storage_search(qs[j], k)
        assert np.array_equal(bi[j], si[0]) and np.array_equal(bd[j], sd[0])
    st.close()


def test_c3_full_size_fp16_batch64(ctx):
    n, d, k = 10_000_000, 768, 100
    st = oa.EmbeddingFieldStorage(ctx, dimensions=d, reserve_rows=n + 64, dtype=N.DTYPE_F16)
    st.fill_synthetic(n, seed=0xC0FFEE)
    qs = util.gaussian_rows(64, d, seed=0xBEEF)
    _check_store(st, n, d, k, qs[0], quantised=True)
    ids, dist, cnt = st.storage_search(qs, k)
    assert cnt.tolist() == [k] * 64 and np.all(np.diff(dist, axis=1) >= 0)
    for j in (0, 31, 63):  # batch == solo, and distances hold on the stored rows
        si, sd, sc = st.storage_search(qs[j], k)
        assert np.array_equal(ids[j], si[0]) and np.array_equal(dist[j], sd[0])
        rows, _ = st.get_rows(ids[j])
        assert np.max(np.abs(orc.distances(rows, qs[j]) - dist[j])) <= TOL
    st.close()


def test_c4_full_size_bm25_bit_exact(ctx):
    n, T, k = 10_000_000, 12, 100
    rng = np.random.default_rng(0xB26)
    ranks = np.unique(np.exp(rng.uniform(np.log(100), np.log(100000), size=256)).astype(np.uint32))
    post = ft.PostingsStore(ctx)
    post.fill_synthetic(n, ranks, seed=0xB25)
    avg = post.info()["avg_field_length"]
    for trial in range(2):
        ql = rng.choice(len(ranks), size=T, replace=False)
        refs = [(t, int(l), 1.0) for t, l in enumerate(ql)]
        ids, sc, count = post.search(refs, T, float(n), k)
        entries = []
        for t, l in enumerate(ql):
            d_, tf, ln = post.get_list(int(l))
            ntf = (tf.astype(np.float32) / (np.float32(0.25) + np.float32(0.75) * (ln.astype(np.float32) / np.float32(avg)))
                   ).astype(np.float32)
            entries.append((t, d_, np.float32(1.0) * ntf))
        od, os_ = orc.search_full_text(entries, T, float(n), 1.2, None)
        td, ts = orc.top_n(od, os_, k)
        assert count == len(od)
        assert ids.tolist() == td.tolist()
        assert np.array_equal(sc.view(np.uint32), ts.view(np.uint32))
        ids2, sc2, count2 = post.search(refs, T, float(n), k)                    # idempotence (epoch-stamped scratch)
        assert count2 == count and ids2.tolist() == ids.tolist() and np.array_equal(sc2, sc)
    post.close()
