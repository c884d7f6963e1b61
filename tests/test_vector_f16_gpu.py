"""GPU parity of K2: fp16 storage + batched-query MFMA scan with the fused threshold filter.

The fp16 path is a build-side extension (the reference stores f32 and takes one query — SURVEY F4); its
parity is defined against the f32 restatement evaluated on the fp16-ROUNDED vectors: the kernel computes the
exact cosine of the quantised vectors with f32 accumulation, so the 1e-4 bar applies unchanged.
"""
import numpy as np
import pytest

import oramacore_amd as oa
import util
from oracle import oracle as orc
from oramacore_amd import _native as N

pytestmark = pytest.mark.gpu
TOL = 1e-4


def q16(x):
    return np.ascontiguousarray(np.asarray(x, dtype=np.float32).astype(np.float16).astype(np.float32))


def make_store(ctx, corpus, row_doc=None):
    n, d = corpus.shape
    st = oa.EmbeddingFieldStorage(ctx, dimensions=d, dtype=oa.DTYPE_F16)
    ids = np.arange(n, dtype=np.uint64) if row_doc is None else row_doc
    assert st.insert_rows(ids, corpus) == n
    return st


def check(st, corpus16, queries, k, allow=None, dead_rows=None, what=""):
    ids, dist, cnt = st.storage_search(queries, k, allow)
    for qi in range(queries.shape[0]):
        full = orc.distances(corpus16, q16(queries[qi])).astype(np.float64)
        if dead_rows is not None:
            full[dead_rows] = np.nan
        if allow is not None:
            mask = np.array([allow.contains(int(d)) for d in range(corpus16.shape[0])])
            full[~mask] = np.nan
        m = int(cnt[qi])
        util.assert_topk_sound(ids[qi, :m], dist[qi, :m], full, k, TOL, f"{what} q{qi}")


@pytest.mark.parametrize("d", [384, 768, 1024, 100])
@pytest.mark.parametrize("nq", [1, 5, 33, 64, 70])
def test_head_only_batches(ctx, d, nq):
    """N below the dense head: every reference model dimension + a padded odd one; batch sizes that hit one and
    two MFMA column tiles, a ragged tile and the two-pass (> 64 queries) loop."""
    n = 3000 + d
    corpus = util.gaussian_rows(n, d, seed=d)
    queries = util.gaussian_rows(nq, d, seed=d + nq)
    st = make_store(ctx, corpus)
    check(st, q16(corpus), queries, 100, what=f"d={d} nq={nq}")
    # storage round trip: rows come back as the fp16-rounded values
    rows, docs = st.get_rows(np.array([0, 31, 32, n - 1], dtype=np.uint64))
    assert np.array_equal(rows, q16(corpus[[0, 31, 32, n - 1]]))
    st.close()


def test_filter_path_random_and_adversarial_order(ctx):
    """N above the dense head (131072 rows): the rest goes through the threshold filter.  The second corpus is
    ordered by INCREASING similarity to query 0, so every later row beats the running threshold — the
    worst case for the candidate lists — and the result must still be exact."""
    n, d, k = 200_000, 384, 100
    corpus = util.gaussian_rows(n, d, seed=7)
    queries = util.gaussian_rows(64, d, seed=8)
    st = make_store(ctx, corpus)
    check(st, q16(corpus), queries, k, what="random order")
    check(st, q16(corpus), queries[:3], 7, what="random order small k")
    st.close()
    sim = q16(corpus) @ q16(queries[0]) / np.linalg.norm(q16(corpus), axis=1)
    order = np.argsort(sim, kind="stable")
    corpus2 = corpus[order]
    st = make_store(ctx, corpus2)
    check(st, q16(corpus2), queries[:4], k, what="adversarial order")
    st.close()


def test_deletes_filter_and_compaction(ctx):
    n, d, k = 140_000, 384, 50
    corpus = util.gaussian_rows(n, d, seed=17)
    queries = util.gaussian_rows(6, d, seed=18)
    st = make_store(ctx, corpus)
    dead = np.array([3, 64, 65, 131071, 131072, 139_999], dtype=np.int64)
    for r in dead:
        st.delete(int(r))
    allow = oa.AllowBitmap.from_mask(np.arange(n) % 3 != 0)
    check(st, q16(corpus), queries, k, dead_rows=dead, what="dead")
    check(st, q16(corpus), queries, k, allow=allow, dead_rows=dead, what="dead+filter")
    before = st.storage_search(queries, k)
    st.compact(3)
    assert st.info()["num_rows"] == n - len(dead) and not st.has_pending_ops()
    after = st.storage_search(queries, k)
    assert np.array_equal(before[0], after[0]) and np.allclose(before[1], after[1], atol=1e-6)
    st.close()


def test_matches_f32_store_up_to_quantisation(ctx):
    """Same vectors in an f32 store and an fp16 store: distances agree to fp16 quantisation (~1e-3) and the
    top-10 of well-separated neighbours are identical."""
    n, d = 20_000, 768
    corpus = util.gaussian_rows(n, d, seed=27)
    q = util.gaussian_rows(1, d, seed=28)[0]
    corpus[:10] = (q[None, :] * np.linspace(1.0, 0.3, 10, dtype=np.float32)[:, None]
                   + util.gaussian_rows(10, d, seed=29) * np.linspace(0.05, 2.0, 10, dtype=np.float32)[:, None])
    s32 = oa.EmbeddingFieldStorage(ctx, dimensions=d)
    s32.insert_rows(np.arange(n, dtype=np.uint64), corpus)
    s16 = make_store(ctx, corpus)
    i32, d32, _ = s32.storage_search(q, 10)
    i16, d16, _ = s16.storage_search(q, 10)
    assert np.array_equal(i32, i16)
    assert np.max(np.abs(d32 - d16)) < 2e-3
    s32.close()
    s16.close()


def test_synthetic_fill_f16_1m(ctx):
    """1 M x 768 fp16 generated in HBM, batch of 64: planted copies on top, sorted, exact on read-back rows."""
    n, d, k = 1_000_000, 768, 100
    st = oa.EmbeddingFieldStorage(ctx, dimensions=d, dtype=oa.DTYPE_F16, reserve_rows=n + 64)
    st.fill_synthetic(n, seed=0xC0FFEE)
    queries = util.gaussian_rows(64, d, seed=0xBEEF)
    st.insert_rows(np.arange(n, n + 64, dtype=np.uint64), queries * np.float32(1.7))
    ids, dist, cnt = st.storage_search(queries, k)
    assert np.all(cnt == k)
    for qi in range(64):
        assert ids[qi, 0] == n + qi and abs(dist[qi, 0]) < 2e-4
        assert np.all(np.diff(dist[qi]) >= 0)
    rows, _ = st.get_rows(ids[5])
    od = orc.distances(rows, q16(queries[5]))
    assert np.max(np.abs(od - dist[5])) <= TOL
    st.close()


@pytest.mark.parametrize("d", [768, 384, 1024])
def test_wide_batches_equal_solo_queries(ctx, d):
    """K2c (65..256 queries per corpus pass, GEMM-tiled) must return per query exactly what K2 returns for the
    query alone: same ids, bit-identical distances — across the dense head, the threshold-filter super-chunks,
    tombstones and an allow bitmap; and sound against the oracle on the quantised rows."""
    n = 140_000 if d == 768 else 40_000          # 768: beyond the 131072-row dense head → filter path too
    corpus = util.gaussian_rows(n, d, seed=50 + d)
    st = make_store(ctx, corpus, row_doc=np.arange(n, dtype=np.uint64) * 2 + 1)
    for doc in (1, 3, 2 * 77 + 1):
        st.delete(doc)
    bm = oa.AllowBitmap.from_mask((np.arange(2 * n + 2) % 7) != 3)
    queries = util.gaussian_rows(256, d, seed=60 + d)
    picks = (0, 31, 32, 63, 64, 69, 96, 100, 127, 128, 160, 199, 224, 255)
    solos = {id(allow): {i: st.storage_search(queries[i], 30, allow) for i in picks} for allow in (None, bm)}
    # every wide form: 4 = K2q (queries stationary in registers, the default), 2 / 3 = K2d geometries; 5 = K2h (K loop split
    # over a wave pair) and 1 = K2c exist in comparison builds only (ORAMA_COMPARISON_KERNELS=1) — the product library
    # refuses them with ORAMA_ERR_UNSUPPORTED
    from oramacore_amd import _build

    for mode in (4, 5, 2, 3, 1):
        if mode in (1, 5) and not _build.comparison_build():
            with pytest.raises(oa.OramaError) as ei:
                ctx.set_f16_wide(mode)
            assert ei.value.status == N.ORAMA_ERR_UNSUPPORTED
            continue
        ctx.set_f16_wide(mode)
        # repeated: the pipelines are asynchronous (LDS DMA rings) — catch races
        for allow in ((None, bm, None, None) if mode == 4 else (None, bm)):
            solo = solos[id(allow)]
            for nq in ((65, 70, 128, 129, 200, 256) if mode == 4 else (70, 128, 256)):
                ids, dist, cnt = st.storage_search(queries[:nq], 30, allow)
                assert cnt.tolist() == [30] * nq
                for i in picks:
                    if i >= nq:
                        continue
                    assert np.array_equal(ids[i], solo[i][0][0]), (mode, nq, i)
                    assert np.array_equal(dist[i].view(np.uint32), solo[i][1][0].view(np.uint32)), (mode, nq, i)
    ctx.set_f16_wide(4)
    c16 = q16(corpus)
    ids, dist, cnt = st.storage_search(queries[:70], 30)
    for qi in (0, 69):
        full = orc.distances(c16, q16(queries[qi])).astype(np.float64)
        full[[0, 1, 77]] = np.nan
        util.assert_topk_sound((ids[qi] - 1) // 2, dist[qi], full, 30, TOL, f"wide q{qi}")
    st.close()


@pytest.mark.parametrize("nq", [1, 40, 70, 200])
def test_l2_metric_on_fp16_storage(ctx, nq):
    """Squared-L2 on the fp16 store (north_star: cosine/L2): the MFMA kernels accumulate q.x and finish with
    (|q|^2 + |x|^2) - 2 q.x, norms of the fp16-rounded vectors in f32.  Against the oracle's direct sum of squared
    differences on the same rounded vectors; the expanded form cancels ~|q|^2 + |x|^2 (<= 8 here), so the absolute
    bar is 3e-4 instead of the cosine path's 1e-4.  Sizes cross the dense head so K2 (<= 64), K2d (65..256) and the
    threshold filter all run in L2 mode."""
    n, d, k = 140_000, 384, 50
    corpus = util.gaussian_rows(n, d, seed=31)
    queries = util.gaussian_rows(nq, d, seed=32)
    st = oa.EmbeddingFieldStorage(ctx, dimensions=d, dtype=oa.DTYPE_F16, metric=oa.METRIC_L2SQ)
    assert st.insert_rows(np.arange(n, dtype=np.uint64), corpus) == n
    c16 = q16(corpus)
    ids, dist, cnt = st.storage_search(queries, k)
    for qi in sorted({0, nq // 2, nq - 1}):
        full = orc.distances(c16, q16(queries[qi]), metric=1).astype(np.float64)
        util.assert_topk_sound(ids[qi, :cnt[qi]], dist[qi, :cnt[qi]], full, k, 3e-4, f"l2 f16 q{qi}")
    st.close()


def _crowd(n, d, nq, seed):
    """A corpus ordered by INCREASING similarity to a crowd of near-identical queries: in the filter region every
    row beats every query's running threshold, so a 32-row tile hands 32 x nq rows to the staging area."""
    rng = np.random.default_rng(seed)
    centre = rng.standard_normal(d).astype(np.float32)
    queries = (centre[None, :] + 0.01 * rng.standard_normal((nq, d))).astype(np.float32)
    corpus = util.gaussian_rows(n, d, seed=seed + 1)
    c16 = q16(corpus)
    sim = c16 @ q16(centre) / np.linalg.norm(c16, axis=1)
    return corpus[np.argsort(sim, kind="stable")], queries


@pytest.mark.parametrize("d,nq", [(64, 64), (1024, 64), (2048, 40), (768, 33)])
def test_filter_path_every_row_passes_for_every_query(ctx, d, nq):
    """The worst case for the per-wave staging of passing rows: full tiles of candidates for all
    queries at once, at dimensions where the staging area is large (64: 1 024 rows per wave), the smallest it gets
    with two column tiles (1 024: 192 rows) and where 64 queries no longer fit one pass (2 048: passes of 32)."""
    n = 131_072 + 6_000
    corpus, queries = _crowd(n, d, nq, seed=900 + d)
    st = make_store(ctx, corpus)
    check(st, q16(corpus), queries, 100, what=f"crowd d={d} nq={nq}")
    st.close()


def test_filter_path_crowd_with_deletes_and_allow_bitmap(ctx):
    """Same crowd, with tombstones and an allow bitmap in the filter region: the bitmap lookup happens when the
    staged rows are flushed, the tombstone test when they are staged."""
    n, d, nq = 131_072 + 9_000, 128, 64
    corpus, queries = _crowd(n, d, nq, seed=77)
    st = make_store(ctx, corpus)
    dead = np.arange(131_072 + 5, n, 97, dtype=np.int64)
    for r in dead:
        st.delete(int(r))
    mask = np.arange(n) % 5 != 1
    allow = oa.AllowBitmap.from_mask(mask)
    ids, dist, cnt = st.storage_search(queries, 50, allow)
    c16 = q16(corpus)
    for qi in (0, 1, 31, 32, 47, 63):
        full = orc.distances(c16, q16(queries[qi])).astype(np.float64)
        full[dead] = np.nan
        full[~mask] = np.nan
        m = int(cnt[qi])
        util.assert_topk_sound(ids[qi, :m], dist[qi, :m], full, 50, TOL, f"crowd dead+filter q{qi}")
    st.close()


_CHUNK_SCRIPT = r"""
import hashlib, json, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
import oramacore_amd as oa
ctx = oa.Context(0)
for opt in sys.argv[2:]:
    name, value = opt.split("=")
    ctx.set_option(name, int(value))
n, d, k = 1_300_000, 64, 50
st = oa.EmbeddingFieldStorage(ctx, dimensions=d, dtype=oa.DTYPE_F16, reserve_rows=n)
st.fill_synthetic(n, seed=0xABCD)
rng = np.random.default_rng(5)
out = {}
for nq in (3, 64, 70):
    q = rng.standard_normal((nq, d)).astype(np.float32)
    ids, dist, cnt = st.storage_search(q, k)
    h = hashlib.sha256()
    h.update(ids.tobytes()); h.update(dist.tobytes()); h.update(cnt.tobytes())
    out[str(nq)] = h.hexdigest()
print(json.dumps(out))
"""


def test_super_chunks_and_growing_chunks_give_the_same_answer(tmp_path):
    """The filter scan runs in super-chunks (candidate budget) that may grow geometrically; both are context options
    (orama_ctx_set_option "f16_cand_mib" / "f16_chunk_grow"); each variant runs in its own process: one chunk (default budget),
    two chunks of >= 1 M rows (1 MiB budget), and growing chunks (131 072, 262 144, ... rows) — for K2 (3 and 64
    queries) and the wide kernel (70).  Ids, distance bits and counts must be identical."""
    import json, os, subprocess, sys
    from pathlib import Path
    root = str(Path(__file__).resolve().parent.parent)
    script = tmp_path / "chunks.py"
    script.write_text(_CHUNK_SCRIPT)
    outs = []
    for extra in ([], ["f16_cand_mib=1"], ["f16_cand_mib=1", "f16_chunk_grow=1"]):
        r = subprocess.run([sys.executable, str(script), root, *extra], env=dict(os.environ), capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(json.loads(r.stdout.strip().splitlines()[-1]))
    assert outs[0] == outs[1] == outs[2], outs
