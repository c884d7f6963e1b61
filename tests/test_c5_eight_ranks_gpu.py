"""BASELINE configs[4] as the 8-RANK JOB it is (VERDICT r03 "missing" #1): `python bench.py --gpus 8 --workload c5`
— eight processes, 10 M x 768 fp16 rows each (8 x 15.4 GB + scratch fit the one 288 GB GPU of this box), 256 queries per
batch, one all-gather of the per-shard candidates + K6 per batch.  The box has one GPU and real RCCL refuses two ranks on
one device, so the bytes travel over tests/mock_rccl (ORAMA_RCCL_LIB); everything above the transport is the product:
bench.py's own launcher, its preflight, orama_shard_group_create_rank / ncclCommInitRank in eight processes, the pipelined
session, `ranks_seen` in the JSON line.

Bar: the job finishes with one JSON line naming 8 ranks (8 distinct processes); every rank holds the SAME answer, bit for
bit; that answer equals the merge of the eight shards searched one after the other in THIS process (ids, distance bits,
counts — the tie rule included); the size-independent properties hold on it (complete, sorted, no sampled row of any
shard beats the reported k-th distance).
"""
import json
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

import oramacore_amd as oa
from oracle import oracle as orc
from oramacore_amd import _native as N

pytestmark = pytest.mark.gpu

ROOT = Path(__file__).resolve().parent.parent
MOCK = ROOT / "tests" / "mock_rccl" / "libmock_rccl.so"
G, N_TOTAL, D, K, Q = 8, 80_000_000, 768, 100, 256
TOL = 1e-4


def test_c5_eight_rank_job_over_the_loopback_transport(ctx, tmp_path):
    if ctx.device_info()["hbm_bytes"] < 250 * 2**30:
        pytest.skip("needs ~190 GB of HBM for eight 10 M-row shards on one GPU")
    if not MOCK.exists():
        subprocess.run(["make", "-C", str(MOCK.parent)], check=True, capture_output=True)
    env = dict(os.environ, ORAMA_RCCL_LIB=str(MOCK), HSA_ENABLE_IPC_MODE_LEGACY="0")
    for v in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(v, None)
    dump = tmp_path / "c5"
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", str(G), "--workload", "c5", "--steps", "3", "--warmup", "1",
                        "--no-cpu-baseline", "--no-pmc", "--dump-result", str(dump),
                        "--details-file", str(tmp_path / "details.json")],
                       env=env, capture_output=True, text=True, timeout=1500, cwd=str(ROOT))
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    last = r.stdout.strip().splitlines()[-1]
    assert len(last) < 4096, len(last)  # the metric line of an N > 1 job obeys the same budget as the N = 1 one
    compact = json.loads(last)
    assert "error" not in compact, compact
    assert compact["n_gpus"] == G and compact["config"]["ranks_seen"] == G and compact["config"]["comm_world"] == G
    assert compact["roofline"]["frac"] > 0 and compact["value"] > 0
    line = json.loads((tmp_path / "details.json").read_text())  # the long form: who was in the job, spans, samples
    assert abs(line["value"] - compact["value"]) / line["value"] < 1e-5
    assert line["n_gpus"] == G and line["config"]["comm_world"] == G and line["config"]["valid"] is True
    assert line["config"]["rows_total"] == N_TOTAL and line["config"]["rows_per_gpu"] == N_TOTAL // G
    seen = line["config"]["ranks_seen"]
    assert [e["rank"] for e in seen] == list(range(G)) and len({e["pid"] for e in seen}) == G, seen
    assert line["value"] > 0 and line["config"]["queries_per_step"] == Q
    # BASELINE.json's metric names p50 latency at every N: the 8-rank line carries it (slowest rank per step), with the HIP-event
    # spans of the step's parts — all-gather and K6 among them
    assert line["latency_ms_p50"] > 0 and line["latency_ms_p95"] >= line["latency_ms_p50"] and line["latency_samples"] >= 5
    bd = line["step_breakdown_us"]
    assert bd["scan"] > 0 and bd["all_gather"] is not None and bd["merge_k6"] is not None

    ranks = [np.load(f"{dump}.rank{g}.npz") for g in range(G)]
    ids, dist, cnt, qs = ranks[0]["ids"], ranks[0]["dist"], ranks[0]["cnt"], ranks[0]["queries"]
    assert ids.shape == (Q, K) and cnt.tolist() == [K] * Q and np.all(np.diff(dist, axis=1) >= 0)
    for g in range(1, G):  # every rank holds the global answer
        assert np.array_equal(ranks[g]["ids"], ids) and np.array_equal(ranks[g]["dist"].view(np.uint32), dist.view(np.uint32)), g
        assert np.array_equal(ranks[g]["cnt"], cnt) and np.array_equal(ranks[g]["queries"], qs), g
        assert int(ranks[g]["lo"]) == g * (N_TOTAL // G) and int(ranks[g]["hi"]) == (g + 1) * (N_TOTAL // G)

    # the eight shards one after the other in this process + a host merge by (distance, id): the same bits
    qh = qs.astype(np.float16).astype(np.float32)  # what the fp16 path scores
    cand_ids, cand_dist = [], []
    for g in range(G):
        lo = g * (N_TOTAL // G)
        st = oa.EmbeddingFieldStorage(ctx, dimensions=D, reserve_rows=N_TOTAL // G, dtype=N.DTYPE_F16)
        st.fill_synthetic(N_TOTAL // G, seed=0xC0FFEE + g, first_doc_id=lo)
        si, sd, sc = st.storage_search(qs, K)
        assert sc.tolist() == [K] * Q
        cand_ids.append(si)
        cand_dist.append(sd)
        # the reported distances are the oracle's on the rows read back, and no sampled row of this shard beats the k-th
        for j in (0, 131, 255):
            mine = (ids[j] >= lo) & (ids[j] < lo + N_TOTAL // G)
            if mine.any():
                rows, docs = st.get_rows((ids[j][mine] - np.uint64(lo)).astype(np.uint64))
                assert np.array_equal(docs, ids[j][mine])
                assert np.max(np.abs(orc.distances(rows, qh[j]) - dist[j][mine])) <= TOL, (g, j)
            sample = np.random.default_rng(1000 * g + j).choice(N_TOTAL // G, size=20_000, replace=False).astype(np.uint64)
            srows, sdocs = st.get_rows(sample)
            sd_ = orc.distances(srows, qh[j], threads=8)
            inside = set(ids[j].tolist())
            bad = [int(d) for d, x in zip(sdocs.tolist(), sd_.tolist()) if x < dist[j][-1] - 2 * TOL and int(d) not in inside]
            assert not bad, (g, j, bad[:5])
        st.close()
    all_ids = np.concatenate(cand_ids, axis=1)
    all_dist = np.concatenate(cand_dist, axis=1)
    for j in range(Q):
        order = np.lexsort((all_ids[j], all_dist[j]))[:K]  # distance ascending, then DocumentId ascending
        assert np.array_equal(all_ids[j][order], ids[j]), j
        assert np.array_equal(all_dist[j][order].view(np.uint32), dist[j].view(np.uint32)), j
