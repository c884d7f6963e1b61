#!/bin/bash
# Round 4, GPU call 5: 16-ary bounds search, rank-form sharded full-text batches (multirank tests + rate), driver bench,
# native serving figures.
set +e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
O=$R/gpurun_out/r04c
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
echo "== pytest K3r-facing + sharded tests first"
timeout 1200 python -m pytest tests/test_bm25_ranges_gpu.py tests/test_fulltext_gpu.py tests/test_random_gpu.py tests/test_sharded_fulltext_gpu.py tests/test_shard_group_gpu.py tests/test_multirank_gpu.py -q -m gpu -p no:cacheprovider -x 2>&1 | tail -25 | tee $O/pytest_k3r_shard.log
echo "== k3r chunk probe"
timeout 200 python scripts/k3r_chunk_probe.py 2>&1 | tail -1 | tee $O/k3r_probe.log
echo "== shard batch: co-located and rank form"
timeout 400 python scripts/bench_shard_post_batch.py 2>&1 | tail -4 | tee $O/shard_post_batch.log
timeout 400 python scripts/bench_shard_post_batch_ranks.py 2 2>&1 | grep "ranks, one process" | tee $O/shard_post_batch_ranks2.log
timeout 400 python scripts/bench_shard_post_batch_ranks.py 4 2>&1 | grep "ranks, one process" | tee $O/shard_post_batch_ranks4.log
echo "== pytest -m gpu (all)"
timeout 2400 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -12 | tee $O/pytest_gpu.log
echo "== bench (driver command)"
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_all.json 2> $O/bench_all.err; tail -c 300 $O/bench_all.json; echo; tail -5 $O/bench_all.err
echo "== native serving bm25"
true
ls scripts/native
timeout 300 scripts/native/bench_serving bm25 2>&1 | tail -12 | tee $O/serving_bm25.log
du -sh $O
