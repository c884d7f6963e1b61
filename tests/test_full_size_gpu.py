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
    # a batch of 5 equals the solo answers — through K1x (round 6: 2..8 queries over >= 4 GB of rows take the convert-in-registers
    # candidate scan too) and through K1b (the matrix path off)
    qs = util.gaussian_rows(5, d, seed=77)
    bi, bd, bc = st.storage_search(qs, k)
    ctx.set_f32_batch(0)
    try:
        ki, kd, kc = st.storage_search(qs, k)
        solo = {j: st.storage_search(qs[j], k) for j in (0, 4)}
    finally:
        ctx.set_f32_batch(9)
    assert np.array_equal(bi, ki) and np.array_equal(bd.view(np.uint32), kd.view(np.uint32)) and np.array_equal(bc, kc)
    for j, (si, sd, sc) in solo.items():
        assert np.array_equal(bi[j], si[0]) and np.array_equal(bd[j], sd[0])
    # the north-star rows asked 64 + 6 queries at a time (K1x: two passes; K1m with the option): the single-query scan's bits
    # for every query looked at, under a filter as well, and the plan really ran on the matrix cores
    qb = util.gaussian_rows(70, d, seed=78)
    bm = oa.AllowBitmap.from_mask(np.arange(n + 3) % 3 != 0).to_device(ctx)
    for option in (1, 0):
        ctx.set_option("f32_batch_cvt", option)
        ctx.prof_reset(); ctx.prof_enable(True)
        bi, bd, bc = st.storage_search(qb, k)
        fi, fd, fc = st.storage_search(qb[:12], k, bm)
        ctx.prof_enable(False)
        assert ctx.prof_get("vec_scan_f32_cvt" if option else "vec_scan_f32_mfma")[1] >= 4
        assert ctx.prof_get("vec_scan_f32")[1] == 0, "a query fell back to the plain scan: the candidate lists were not proven"
        ctx.set_f32_batch(0)
        try:
            for j in (0, 31, 32, 63, 64, 69):
                si, sd, sc = st.storage_search(qb[j], k)
                assert np.array_equal(bi[j], si[0]) and np.array_equal(bd[j].view(np.uint32), sd[0].view(np.uint32)) and bc[j] == sc[0], (option, j)
            for j in (0, 11):
                si, sd, sc = st.storage_search(qb[j], k, bm)
                assert np.array_equal(fi[j], si[0]) and np.array_equal(fd[j].view(np.uint32), sd[0].view(np.uint32)) and fc[j] == sc[0], (option, j)
        finally:
            ctx.set_f32_batch(9)
    ctx.set_option("f32_batch_cvt", 1)
    bm.close()
    st.close()


def test_ns_full_size_two_stage_equals_the_fp32_scan(ctx):
    """NS at full size through the two-stage plan (fp32 rows + fp16 shadow, DTYPE_F32_SHADOW16): the store passes the same
    property checks as the plain fp32 store, and its answers equal the plain scan's of the SAME store (plan switched off)
    bit for bit — solo, small batch (K1b on the plain side) and a 64-query batch, with and without a filter."""
    n, d, k = 10_000_000, 768, 100
    st = oa.EmbeddingFieldStorage(ctx, dimensions=d, reserve_rows=n + 16, dtype=oa.DTYPE_F32_SHADOW16)
    st.fill_synthetic(n, seed=0xC0FFEE)
    q = util.gaussian_rows(1, d, seed=0xBEEF)[0]
    _check_store(st, n, d, k, q, quantised=False)
    qs = util.gaussian_rows(64, d, seed=99)
    bm = oa.AllowBitmap.from_mask(np.arange(n + 3) % 3 != 0).to_device(ctx)
    two = {"solo": st.storage_search(qs[0], k), "b5": st.storage_search(qs[:5], k), "b64": st.storage_search(qs, k),
           "filtered": st.storage_search(qs[:3], k, bm)}
    used = st.info()
    assert used["two_stage_queries"] >= 73 and used["two_stage_fallbacks"] == 0
    ctx.set_two_stage(False)
    try:
        one = {"solo": st.storage_search(qs[0], k), "b5": st.storage_search(qs[:5], k), "b64": st.storage_search(qs[:16], k),
               "filtered": st.storage_search(qs[:3], k, bm)}
    finally:
        ctx.set_two_stage(True)
    for name in two:
        m = one[name][0].shape[0]
        assert np.array_equal(two[name][0][:m], one[name][0]), name
        assert np.array_equal(two[name][1][:m].view(np.uint32), one[name][1].view(np.uint32)), name
        assert np.array_equal(two[name][2][:m], one[name][2]), name
    bm.close()
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


def _sample_property(st, ids_row, dist_row, q, n, seed, size=50_000):
    """No row of a random sample is better than the reported k-th distance unless it is in the result."""
    rng = np.random.default_rng(seed)
    sample = rng.choice(n, size=size, replace=False).astype(np.uint64)
    srows, _ = st.get_rows(sample)
    sd = orc.distances(srows, q, threads=8)
    inside = set(ids_row.tolist())
    worse = sd >= dist_row[-1] - 2 * TOL
    bad = [int(r) for w, r in zip(worse.tolist(), sample.tolist()) if not w and int(r) not in inside]
    assert not bad, f"rows clearly inside the top-k are missing: {bad[:5]}"


def test_c5_per_gpu_shard_fp16_wide_batch256(ctx):
    """BASELINE configs[4], the shape ONE of the 8 GPUs runs: 10 M x 768 fp16 rows, a batch of 256 queries in one
    corpus pass (K2c: global->LDS DMA ring + register-blocked MFMA tiles), per-query top-100, then the exchange
    tail of the sharded path — packed candidate blocks of 8 (emulated) shards -> K6 merge.

    Checks: sorted + complete; reported distances equal the oracle's on the rows read back (1e-4, the oracle scores
    the fp16-rounded query against the stored fp16 rows); the 50 K-row sample property; planted exact matches on
    top; THREE repetitions bit-identical (the DMA ring has no ordering slack to hide a race behind); wide batch ==
    solo queries (K2); shard/merge identity through the packed C-ABI entry points."""
    n, d, k, Q = 10_000_000, 768, 100, 256
    lib = N.load()
    st = oa.EmbeddingFieldStorage(ctx, dimensions=d, reserve_rows=n + 64, dtype=N.DTYPE_F16)
    st.fill_synthetic(n, seed=0xC0FFEE)
    qs = util.gaussian_rows(Q, d, seed=0xBEEF)
    plant = np.stack([qs[0] * np.float32(s) for s in (0.5, 1.0, 3.0)] + [qs[255] * np.float32(2.0)])
    st.insert_rows(np.arange(n, n + 4, dtype=np.uint64), plant)
    n_all = n + 4
    ids, dist, cnt = st.storage_search(qs, k)
    assert cnt.tolist() == [k] * Q and np.all(np.diff(dist, axis=1) >= 0)
    assert sorted(ids[0, :3].tolist()) == [n, n + 1, n + 2] and np.all(np.abs(dist[0, :3]) < 2e-3)
    assert ids[255, 0] == n + 3 and abs(dist[255, 0]) < 2e-3
    for rep in range(3):                                                  # pipeline races would show here
        ids2, dist2, cnt2 = st.storage_search(qs, k)
        assert np.array_equal(ids, ids2) and np.array_equal(dist, dist2), f"repetition {rep} differs"
    qh = qs.astype(np.float16).astype(np.float32)                        # what the fp16 path scores
    for j in (0, 1, 63, 64, 127, 128, 200, 255):
        rows, docs = st.get_rows(ids[j])
        assert np.array_equal(docs, ids[j])
        assert np.max(np.abs(orc.distances(rows, qh[j]) - dist[j])) <= TOL, j
    for j in (5, 130, 254):
        _sample_property(st, ids[j], dist[j], qh[j], n, seed=100 + j)
    for j in (0, 77, 255):                                                # K2c batch == K2 solo, bit for bit
        si, sd, sc = st.storage_search(qs[j], k)
        assert np.array_equal(ids[j], si[0]) and np.array_equal(dist[j], sd[0]), j

    # ---- the exchange tail of the sharded path through the C ABI: 8 shards emulated by doc-id residues, each
    # writes its packed block [q*k ids][q*k dist] into its slot of the "all-gathered" buffer, K6 merges them
    G = 8
    nb = lib.orama_packed_block_bytes(Q, k)
    assert nb == (Q * k * 12 + 7) // 8 * 8
    d_q = oa.DeviceBuffer(ctx, qs.nbytes).upload(qs)
    d_blocks = oa.DeviceBuffer(ctx, G * nb)
    d_n = oa.DeviceBuffer(ctx, Q * 4)
    d_oi, d_od, d_on = oa.DeviceBuffer(ctx, Q * k * 8), oa.DeviceBuffer(ctx, Q * k * 4), oa.DeviceBuffer(ctx, Q * 4)
    for g in range(G):
        bm = oa.AllowBitmap.from_mask(np.arange(n_all) % G == g).to_device(ctx)
        tok, bits = bm.ffi_args()
        N.check(lib.orama_vec_search_packed_device(st.handle, d_q.ptr, Q, k, tok, bits, d_blocks.ptr + g * nb,
                                                   d_n.ptr, None))
        ctx.synchronize()
        assert d_n.download(np.uint32, Q).tolist() == [k] * Q
        bm.close()
    N.check(lib.orama_merge_packed_device(ctx.handle, d_blocks.ptr, G, Q, k, d_oi.ptr, d_od.ptr, d_on.ptr, None))
    ctx.synchronize()
    m_ids = d_oi.download(np.uint64, Q * k).reshape(Q, k)
    m_dist = d_od.download(np.float32, Q * k).reshape(Q, k)
    assert d_on.download(np.uint32, Q).tolist() == [k] * Q
    assert np.array_equal(m_ids, ids) and np.array_equal(m_dist, dist)
    for b in (d_q, d_blocks, d_n, d_oi, d_od, d_on):
        b.free()
    st.close()


def test_c4_full_size_hybrid_bit_exact(ctx):
    """BASELINE configs[3] at full size through ONE call (orama_hybrid_search: vector leg and BM25F leg on two HIP
    streams, a2 epilogue inside the library, K5 combine, OMC, count, K4): 10 M documents, 12-token queries
    (~600 K postings) + 10 M x 768 fp32 rows.  The result must equal the oracle pipeline bit for bit GIVEN the
    device's vector hits (whose distances are themselves checked against the oracle on the stored rows, 1e-4, and
    by the sample property): normalize_and_combine -> apply_omc -> top_n, ids / scores / count."""
    n, d, k, T = 10_000_000, 768, 100, 12
    vec = oa.EmbeddingFieldStorage(ctx, dimensions=d, reserve_rows=n)
    vec.fill_synthetic(n, seed=0xC0FFEE)
    rng = np.random.default_rng(0xB26)
    ranks = np.unique(np.exp(rng.uniform(np.log(100), np.log(100000), size=256)).astype(np.uint32))
    post = ft.PostingsStore(ctx)
    post.fill_synthetic(n, ranks, seed=0xB25)
    avg = np.float32(post.info()["avg_field_length"])
    omc_docs = rng.choice(n, size=2000, replace=False).astype(np.uint64)
    omc = {int(dd): float(m) for dd, m in zip(omc_docs, rng.choice([0.25, 0.5, 2.0, 5.0, 10.0], size=2000))}
    post.set_omc(omc)
    qv = util.gaussian_rows(3, d, seed=0xBEEF)
    for trial in range(3):
        ql = rng.choice(len(ranks), size=T, replace=False)
        refs = [(t, int(l), 1.0) for t, l in enumerate(ql)]
        apply_omc = trial != 0
        sim_min = 0.0 if trial < 2 else 0.02
        h_ids, h_sc, h_count = post.hybrid_search(vec, qv[trial], k, sim_min, refs, T, float(n), k,
                                                  apply_omc=apply_omc)
        # oracle: full-text map from the stored postings
        entries = []
        for t, l in enumerate(ql):
            d_, tf, ln = post.get_list(int(l))
            ntf = (tf.astype(np.float32) / (np.float32(0.25) + np.float32(0.75) * (ln.astype(np.float32) / avg))
                   ).astype(np.float32)
            entries.append((t, d_, np.float32(1.0) * ntf))
        od, os_ = orc.search_full_text(entries, T, float(n), 1.2, None)
        # vector leg: the device's hits, validated against the oracle on the rows read back
        v_ids, v_dist, v_cnt = vec.storage_search(qv[trial], k)
        assert v_cnt[0] == k
        rows, _ = vec.get_rows(v_ids[0])
        assert np.max(np.abs(orc.distances(rows, qv[trial]) - v_dist[0])) <= TOL
        _sample_property(vec, v_ids[0], v_dist[0], qv[trial], n, seed=7 + trial, size=20_000)
        vmap = orc.embedding_epilogue(v_ids[0], v_dist[0], False, sim_min)
        cd, cs = orc.normalize_and_combine(list(vmap), list(vmap.values()), od, os_)
        if apply_omc:
            cs = orc.apply_omc(cd, cs, list(omc), list(omc.values()))
        td, ts = orc.top_n(cd, cs, k)
        assert h_count == len(cd), trial
        assert h_ids.tolist() == td.tolist(), trial
        assert np.array_equal(h_sc.view(np.uint32), ts.view(np.uint32)), trial
    post.close()
    vec.close()


def test_two_stage_candidates_crowded_into_one_wave(ctx):
    """The one-query candidate stage keeps 64 rows per wave.  70 near-copies of the query are planted 131 072 rows apart —
    the stride at which the same wave of the 2 048-wave grid comes round again — so one wave must evict rows that
    belong to the answer: the stage has to notice (wave_thr check) and the plan has to fall back to the fp32 scan."""
    d, k, stride, n_plant = 384, 10, 131072, 70
    rng = np.random.default_rng(41)
    q = rng.standard_normal(d).astype(np.float32)
    plant = (q[None, :] + rng.standard_normal((n_plant, d)).astype(np.float32) * np.float32(0.02)).astype(np.float32)
    stores = []
    for dt in (oa.DTYPE_F32, oa.DTYPE_F32_SHADOW16):
        st = oa.EmbeddingFieldStorage(ctx, dimensions=d, reserve_rows=n_plant * stride + 16, dtype=dt)
        for i in range(n_plant):
            st.fill_synthetic(stride - 1, seed=1000 + i, first_doc_id=i * stride)
            st.insert_rows(np.array([i * stride + stride - 1], dtype=np.uint64), plant[i:i + 1])
        assert st.info()["num_rows"] == n_plant * stride
        stores.append(st)
    plain, shadow = stores
    for kk in (k, 64):
        a = plain.storage_search(q, kk)
        b = shadow.storage_search(q, kk)
        assert a[2].tolist() == b[2].tolist() and a[0].tolist() == b[0].tolist()
        assert np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32))
        assert set((a[0][0] % stride).tolist()) == {stride - 1}  # the planted rows are the answer
    info = shadow.info()
    assert info["two_stage_queries"] == 2 and info["two_stage_fallbacks"] == 2
    # a query that is not crowded takes no fallback
    other = rng.standard_normal(d).astype(np.float32)
    a, b = plain.storage_search(other, 100), shadow.storage_search(other, 100)
    assert a[0].tolist() == b[0].tolist() and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32))
    assert shadow.info()["two_stage_fallbacks"] == 2
    plain.close()
    shadow.close()
