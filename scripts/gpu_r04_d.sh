#!/bin/bash
# Round 4, GPU call 7: tie rule of the final selection, bounds lanes by batch size, k-way merge of the sharded batches, fp16 legs
# with two steps in flight, whole suite, driver bench, serving.
set +e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
O=$R/gpurun_out/r04d
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
echo "== pytest -m gpu (all)"
timeout 2400 python -m pytest tests -q -m gpu -p no:cacheprovider -x 2>&1 | tail -12 | tee $O/pytest_gpu.log
echo "== k3r A/B"
timeout 600 python scripts/k3r_ab.py 2>&1 | tail -16 | tee $O/k3r_ab.log
echo "== shard batch: co-located and rank form"
timeout 400 python scripts/bench_shard_post_batch.py 2>&1 | tail -3 | tee $O/shard_post_batch.log
timeout 400 python scripts/bench_shard_post_batch_ranks.py 2 2>&1 | grep "ranks, one process" | tee $O/shard_post_batch_ranks2.log
echo "== bench (driver command)"
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_all.json 2> $O/bench_all.err; tail -c 300 $O/bench_all.json; echo; tail -5 $O/bench_all.err
echo "== native serving bm25"
timeout 300 scripts/native/bench_serving bm25 2>&1 | tail -9 | tee $O/serving_bm25.log
du -sh $O
