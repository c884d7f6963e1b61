#!/usr/bin/env python3
"""Full-text batches over a shard group whose ranks live in SEPARATE PROCESSES (the shape bench.py runs under
torch.distributed.run): C4-shaped queries, 10 M documents split over WORLD ranks on the one GPU, the loopback transport of
tests/mock_rccl.  orama_shard_post_search_batch = df pass, ONE all-reduce, scoring pass, ONE all-gather per block of 512
queries (round 3: this shape answered a batch one staged query at a time).

    python scripts/bench_shard_post_batch_ranks.py [WORLD]        # the parent: starts WORLD ranks of itself
"""
import os
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
n, T, k = 10_000_000, 12, 100


def rank_main(rank: int, world: int, uid_hex: str):
    import numpy as np

    import oramacore_amd as oa
    from oramacore_amd import fulltext as ft
    from oramacore_amd.shard_group import ShardGroup

    rng = np.random.default_rng(0xB26)
    ranks = np.unique(np.exp(rng.uniform(np.log(100), np.log(100000), size=2048)).astype(np.uint32))
    qs = [([(t, int(l), 1.0) for t, l in enumerate(rng.choice(len(ranks), size=T, replace=False))], T, None) for _ in range(512)]
    g = ShardGroup.from_rank(bytes.fromhex(uid_hex), rank, world, 0)
    per = n // world
    p = ft.PostingsStore(g.ctx(0))
    p.fill_synthetic(per, ranks, seed=0xB25 + rank, first_doc_id=rank * per)
    bm = oa.AllowBitmap.from_mask((np.arange(n) % 7) != 3).to_device(g.ctx(0))
    for tag, allow in (("unfiltered", None), ("NOT-deleted filter, resident (df counted on every shard by the warm-up call, remembered since)", [bm])):
        g.post_search_batch([p], qs, float(n), k, allow=allow)
        g.barrier()
        t0 = time.perf_counter()
        res = g.post_search_batch([p], qs, float(n), k, allow=allow)
        g.barrier()
        el = time.perf_counter() - t0
        if rank == 0:
            print(f"{world} ranks, one process each: {len(qs) / el:9.0f} queries/s  {tag}  (orama_shard_post_search_batch, Python marshalling "
                  f"included; top-1 of query 0: {int(res[0][0][0])} {float(res[0][1][0]):.6f}, count {res[0][2]})", flush=True)
    if rank == 0:
        t0 = time.perf_counter()
        for q in qs[:32]:
            g.post_search([p], q[0], T, float(n), k)
        print(f"{world} ranks, one process each: {32 / (time.perf_counter() - t0):9.0f} queries/s  one by one (orama_shard_post_search, the staged query)",
              flush=True)
    else:
        for q in qs[:32]:
            g.post_search([p], q[0], T, float(n), k)
    g.barrier()
    bm.close()
    p.close()
    g.close()


if __name__ == "__main__":
    if len(sys.argv) >= 4 and sys.argv[1] == "--rank":
        rank_main(int(sys.argv[2]), int(sys.argv[3]), sys.argv[4])
    else:
        world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
        mock = ROOT / "tests" / "mock_rccl" / "libmock_rccl.so"
        if not mock.exists():
            subprocess.run(["make", "-C", str(mock.parent)], check=True, capture_output=True)
        env = dict(os.environ, ORAMA_RCCL_LIB=str(mock), HSA_ENABLE_IPC_MODE_LEGACY="0")
        uid = (f"/orama_mock_bench_{os.getpid()}".encode()).ljust(128, b"\0").hex()
        procs = [subprocess.Popen([sys.executable, __file__, "--rank", str(r), str(world), uid], env=env) for r in range(world)]
        rc = 0
        for p in procs:
            rc = rc or p.wait(timeout=900)
        sys.exit(rc)
