"""One process of a multi-rank shard-group job (tests/test_multirank_gpu.py starts `world` of these on GPU 0).

    python multirank_worker.py <form> <rank> <world> <uid hex | -> <out.npz>

form = "rank"    : this process is ONE rank of `world` (orama_shard_group_create_rank / ncclCommInitRank) — the
                   one-process-per-GPU shape bench.py uses under torch.distributed.run;
form = "initall" : this process holds ALL `world` shards, each with its own communicator (orama_shard_group_create with
                   ORAMA_SHARD_FORCE_RCCL / ncclCommInitAll, collectives inside ncclGroupStart/End) — the shape of the
                   reference's single ReadSide process driving N GPUs.
ORAMA_RCCL_LIB (set by the parent) names tests/mock_rccl/libmock_rccl.so: real RCCL refuses two ranks on one GPU.

Every search goes through the product's C ABI (orama_shard_*); the worker only builds its shards of the deterministic
corpus below and writes what it got.  The parent compares every rank's output with the single-store answer.
"""
from __future__ import annotations

import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
for p in (str(HERE.parent), str(HERE)):
    if p not in sys.path:
        sys.path.insert(0, p)

import oramacore_amd as oa  # noqa: E402
import util  # noqa: E402
from oramacore_amd import _native as N  # noqa: E402
from oramacore_amd import fulltext as ft  # noqa: E402
from oramacore_amd.shard_group import FORCE_RCCL, ShardGroup  # noqa: E402

# ---------------------------------------------------------------- the deterministic job (shared with the parent)
VEC_N, VEC_D = 6000, 384
WIDE_N, WIDE_D = 4096, 768
K = 50


def cuts_for(world: int, n: int) -> list[int]:
    """Ragged contiguous ranges (one of them tiny) — slot_of() and the merge must not assume equal shards."""
    if world == 1:
        return [0, n]
    if world == 2:
        return [0, n * 5 // 12, n]
    inner = [n // 4, n // 4 + 7] + [n // 4 + 7 + (n - n // 4 - 7) * i // (world - 2) for i in range(1, world - 2)]
    return [0] + inner + [n]


def vector_data():
    rows = util.gaussian_rows(VEC_N, VEC_D, seed=31)
    rows[100] = rows[4000]  # equal distances on different ranks: the tie rule must survive the exchange
    doc_ids = np.arange(VEC_N, dtype=np.uint64) * np.uint64(3) + np.uint64(7)
    qs = util.gaussian_rows(5, VEC_D, seed=32)
    qs[3] = rows[4000]
    return rows, doc_ids, qs


def wide_data():
    rows = util.gaussian_rows(WIDE_N, WIDE_D, seed=41)
    doc_ids = np.arange(WIDE_N, dtype=np.uint64)
    qs = util.gaussian_rows(256, WIDE_D, seed=42)
    return rows, doc_ids, qs


def text_data():
    meta = util.load_json("bm25_synth.json")
    fields = util.mg.zipf_corpus(meta["n_docs"], meta["vocab"], meta["n_fields"], seed=meta["seed"])
    doc_ids = np.arange(meta["n_docs"], dtype=np.uint64) * np.uint64(meta["doc_id_mul"]) + np.uint64(meta["doc_id_add"])
    allow = (util.hash_u64(doc_ids + np.uint64(5)) % np.uint64(3)) != 0
    return meta, fields, doc_ids, allow


TEXT_CASES = (0, 5, 12, 13, 24, 30)


def text_vector_map(doc_ids):
    """The (global) vector leg of the hybrid cases: documents spread over every shard, some outside any posting list."""
    rng = np.random.default_rng(77)
    pick = rng.choice(len(doc_ids), size=14, replace=False)
    return {int(doc_ids[i]): float(s) for i, s in zip(pick, rng.uniform(-0.2, 1.0, size=14))}


def build_text_shard(ctx, meta, fields, doc_ids, lo, hi):
    keys = [(f, term) for f in range(meta["n_fields"]) for term in sorted(fields[f]["postings"])]
    list_id = {key: i for i, key in enumerate(keys)}
    avg = [fields[f]["avg"] for f in range(meta["n_fields"])]  # index-wide averages
    lists = []
    for f, term in keys:
        pl = [p for p in fields[f]["postings"][term] if lo <= p[0] < hi]
        dix = np.array([p[0] for p in pl], dtype=np.int64)
        lists.append(ft.PostingList(field=f, docs=doc_ids[dix], tf=np.array([p[1] for p in pl], dtype=np.uint32),
                                    field_len=fields[f]["lens"][dix] if len(dix) else np.zeros(0, np.uint32)))
    st = ft.PostingsStore(ctx)
    st.build(doc_ids[lo:hi], avg, lists)
    return st, list_id


def refs_of(meta, list_id, case):
    return [(ti, list_id[(f, t)], meta["boosts"][f]) for ti, t in enumerate(case["terms"])
            for f in range(meta["n_fields"]) if (f, t) in list_id]


def run_job(group: ShardGroup, shard_ids: list[int], world: int) -> dict:
    """Everything a rank does.  `shard_ids` = global indices of the shards this process holds (in local order)."""
    out = {}
    nl = len(shard_ids)
    assert group.world == world and group.n_local == nl and group.first_rank == shard_ids[0]

    # ---- vectors: fp32 and fp16 stores, solo / small batch; resident filter; k larger than a shard
    rows, doc_ids, qs = vector_data()
    cuts = cuts_for(world, VEC_N)
    for name, dtype in (("f32", N.DTYPE_F32), ("f16", N.DTYPE_F16)):
        stores = []
        for li, g in enumerate(shard_ids):
            st = oa.EmbeddingFieldStorage(group.ctx(li), dimensions=VEC_D, dtype=dtype)
            st.insert_rows(doc_ids[cuts[g]:cuts[g + 1]], rows[cuts[g]:cuts[g + 1]])
            stores.append(st)
        for tag, q in (("solo", qs[0]), ("tie", qs[3]), ("batch", qs)):
            ids, dist, cnt = group.vec_search(stores, q, K)
            out[f"vec_{name}_{tag}_ids"], out[f"vec_{name}_{tag}_dist"], out[f"vec_{name}_{tag}_cnt"] = ids, dist, cnt
        bm = oa.AllowBitmap(int(doc_ids.max()) + 1, doc_ids[(np.arange(VEC_N) % 3) != 1])
        res = [bm.to_device(group.ctx(li)) for li in range(nl)]
        ids, dist, cnt = group.vec_search(stores, qs[1], K, allow=res)
        out[f"vec_{name}_filter_ids"], out[f"vec_{name}_filter_dist"], out[f"vec_{name}_filter_cnt"] = ids, dist, cnt
        ids, dist, cnt = group.vec_search(stores, qs[2], 4096 // world)  # the merge's capacity limit
        out[f"vec_{name}_bigk_ids"], out[f"vec_{name}_bigk_dist"], out[f"vec_{name}_bigk_cnt"] = ids, dist, cnt
        if name == "f32":
            # the pipelined session bench.py times: all-gather + K6 on the tail streams, two slots in flight
            sess = group.session(stores, qs[:4], 1, 20, n_slots=2)
            for i in range(6):
                sess.step(i)
            sess.sync()
            for slot in (0, 1):
                s_ids, s_dist, s_cnt = sess.result(slot)
                out[f"sess_slot{slot}_ids"], out[f"sess_slot{slot}_dist"], out[f"sess_slot{slot}_cnt"] = s_ids, s_dist, s_cnt
            sess.close()
            # the same session over stores that also keep an fp16 copy of their rows: every shard takes the two-stage plan in
            # its device form (candidates from the shadow, decision by the fp32 rows, the fallback decided on the device) and
            # the exchange sees the same blocks; query 4 is one the shadow cannot serve (a zero vector)
            sq = np.concatenate([qs[:4], np.zeros((1, VEC_D), dtype=np.float32)])
            for li in range(nl):
                group.ctx(li).set_two_stage(True, always=True)
            shadows = []
            for li, g in enumerate(shard_ids):
                st = oa.EmbeddingFieldStorage(group.ctx(li), dimensions=VEC_D, dtype=N.DTYPE_F32_SHADOW16)
                st.insert_rows(doc_ids[cuts[g]:cuts[g + 1]], rows[cuts[g]:cuts[g + 1]])
                shadows.append(st)
            got = {}
            for tag, ss in (("plain", stores), ("shadow", shadows)):
                sess = group.session(ss, sq, 1, 20, n_slots=2)
                for i in range(5):
                    sess.step(i)
                    sess.sync()
                    got[(tag, i)] = sess.result(i % 2)
                sess.close()
            for i in range(5):
                a, b2 = got[("plain", i)], got[("shadow", i)]
                assert np.array_equal(a[0], b2[0]) and np.array_equal(a[1].view(np.uint32), b2[1].view(np.uint32)) and \
                    np.array_equal(a[2], b2[2]), ("shadow session differs from the plain one", i)
                out[f"sess_shadow_q{i}_ids"], out[f"sess_shadow_q{i}_dist"] = b2[0], b2[1]
            assert all(st.info()["two_stage_queries"] >= 5 for st in shadows)
            for st in shadows:
                st.close()
        for r in res:
            r.close()
        for st in stores:
            st.close()

    # ---- the C5 shape in small: fp16 rows, 256 queries per pass (K2d), candidates all-gathered as ONE 307 KB block
    rows, doc_ids, qs = wide_data()
    cuts = cuts_for(world, WIDE_N)
    stores = []
    for li, g in enumerate(shard_ids):
        st = oa.EmbeddingFieldStorage(group.ctx(li), dimensions=WIDE_D, dtype=N.DTYPE_F16)
        st.insert_rows(doc_ids[cuts[g]:cuts[g + 1]], rows[cuts[g]:cuts[g + 1]])
        stores.append(st)
    ids, dist, cnt = group.vec_search(stores, qs, 100)
    out["wide_ids"], out["wide_dist"], out["wide_cnt"] = ids, dist, cnt
    for st in stores:
        st.close()

    # ---- full-text and hybrid: df all-reduce SUM, min/max all-reduce MAX, block all-gather + count sum
    meta, fields, doc_ids, allow = text_data()
    cuts = cuts_for(world, meta["n_docs"])
    posts, list_id = [], None
    for li, g in enumerate(shard_ids):
        st, list_id = build_text_shard(group.ctx(li), meta, fields, doc_ids, cuts[g], cuts[g + 1])
        posts.append(st)
    bm = oa.AllowBitmap(int(doc_ids.max()) + 1, doc_ids[allow])
    res = [bm.to_device(group.ctx(li)) for li in range(nl)]
    vec = text_vector_map(doc_ids)
    omc = {int(doc_ids[3]): 2.0, int(doc_ids[-2]): 4.0, int(doc_ids[1000]): 0.5}
    for ci in TEXT_CASES:
        case = meta["cases"][ci]
        refs = refs_of(meta, list_id, case)
        n_tok = len(case["terms"])
        total = float(meta["n_docs"])
        ids, sc, count = group.post_search(posts, refs, n_tok, total, 50, threshold=case["threshold"],
                                           allow=res if case["filter"] else None)
        out[f"text{ci}_ids"], out[f"text{ci}_sc"], out[f"text{ci}_count"] = ids, sc, np.uint64(count)
        ids, sc, count = group.post_search(posts, refs, n_tok, total, 30, threshold=case["threshold"], vector=vec,
                                           apply_omc=False)
        out[f"hyb{ci}_ids"], out[f"hyb{ci}_sc"], out[f"hyb{ci}_count"] = ids, sc, np.uint64(count)
    # ---- the batch entry: the range scorer on every shard in three phases — df (list lengths or the counting launch), ONE
    # all-reduce of the df words, scoring with the index-wide idf, ONE all-gather of the shards' top-k blocks — whatever the
    # shape of the group (round 3: one process per rank fell back to one staged query at a time)
    batch = [(refs_of(meta, list_id, meta["cases"][ci]), len(meta["cases"][ci]["terms"]), meta["cases"][ci]["threshold"]) for ci in TEXT_CASES]
    for bi, (ids, sc, count) in enumerate(group.post_search_batch(posts, batch, float(meta["n_docs"]), 40)):
        out[f"batch{bi}_ids"], out[f"batch{bi}_sc"], out[f"batch{bi}_count"] = ids, sc, np.uint64(count)
    # ... and under the NOT-deleted filter: every shard counts its df (filter + several lists per token), ONE all-reduce sums them
    for bi, (ids, sc, count) in enumerate(group.post_search_batch(posts, batch, float(meta["n_docs"]), 40, allow=res)):
        out[f"batchf{bi}_ids"], out[f"batchf{bi}_sc"], out[f"batchf{bi}_count"] = ids, sc, np.uint64(count)
    # request batchers form their batches from whatever arrived: refused when ranks live in other processes
    try:
        group.post_batcher(posts).close()
        out["batcher_refused"] = np.uint64(0)
    except oa.OramaError:
        out["batcher_refused"] = np.uint64(1)
    for st in posts:
        st.set_omc(omc)
    case = meta["cases"][12]
    ids, sc, count = group.post_search(posts, refs_of(meta, list_id, case), len(case["terms"]), float(meta["n_docs"]), 100)
    out["omc_ids"], out["omc_sc"], out["omc_count"] = ids, sc, np.uint64(count)
    for st in posts:
        st.set_omc({})

    # ---- search_hybrid in ONE call over row shards + posting shards of the same documents
    hrows = util.gaussian_rows(meta["n_docs"], 128, seed=51)
    vstores = []
    for li, g in enumerate(shard_ids):
        st = oa.EmbeddingFieldStorage(group.ctx(li), dimensions=128)
        st.insert_rows(doc_ids[cuts[g]:cuts[g + 1]], hrows[cuts[g]:cuts[g + 1]])
        vstores.append(st)
    hq = hrows[17] + 0.3 * hrows[meta["n_docs"] - 5]
    case = meta["cases"][12]
    refs = refs_of(meta, list_id, case)
    for tag, (limit, sim) in (("a", (10, 0.0)), ("b", (40, 0.05))):
        ids, sc, count = group.hybrid_search(vstores, posts, hq, limit, sim, refs, len(case["terms"]), float(meta["n_docs"]), 30)
        out[f"onecall_{tag}_ids"], out[f"onecall_{tag}_sc"], out[f"onecall_{tag}_count"] = ids, sc, np.uint64(count)
    for r in res:
        r.close()
    for st in posts + vstores:
        st.close()

    # ---- bench.py's bracket: barrier + max over ranks of a host double
    group.barrier()
    out["allreduce_max"] = np.float64(group.allreduce_max(1.0 + shard_ids[0]))
    return out


def main():
    form, rank, world, uid_hex, out_path = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
    if form == "rank":
        group = ShardGroup.from_rank(bytes.fromhex(uid_hex), rank, world, 0)
        shard_ids = [rank]
    else:
        group = ShardGroup([0] * world, flags=FORCE_RCCL)
        shard_ids = list(range(world))
    assert group.uses_rccl
    out = run_job(group, shard_ids, world)
    group.close()
    np.savez(out_path, **out)
    print(f"worker {form} rank {rank}/{world}: {len(out)} arrays")


if __name__ == "__main__":
    main()
