#!/bin/bash
# Round 4, GPU call 2: second form of the sort-free K3r kernel (4 fat waves, per-round-count bodies, posting-driven cell
# allocation): parity, A/B against the merge tree, SQ counters, postings-per-range sweep; whole suite; driver bench.
set +e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
O=$R/gpurun_out/r04b
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
echo "== pytest K3r-facing tests first"
timeout 900 python -m pytest tests/test_bm25_ranges_gpu.py tests/test_fulltext_gpu.py tests/test_random_gpu.py tests/test_post_append_gpu.py tests/test_facets_gpu.py tests/test_token_score_gpu.py -q -m gpu -p no:cacheprovider -x 2>&1 | tail -25 | tee $O/pytest_k3r.log
echo "== k3r A/B (comparison library)"
timeout 600 python scripts/k3r_ab.py 2>&1 | tail -30 | tee $O/k3r_ab.log
echo "== postings per range sweep"
for T in 1280 1536 1792; do echo "target $T"; ORAMA_K3R_TARGET=$T timeout 200 python scripts/k3r_chunk_probe.py 2>&1 | tail -1; done | tee $O/k3r_target_sweep.log
echo "== SQ counters"
bash scripts/k3r_sq_pmc.sh > $O/k3r_sq.txt 2>&1; tail -32 $O/k3r_sq.txt | sort -u
echo "== pytest -m gpu (all)"
timeout 2400 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -12 | tee $O/pytest_gpu.log
echo "== bench (driver command)"
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_all.json 2> $O/bench_all.err; tail -c 300 $O/bench_all.json; echo; tail -5 $O/bench_all.err
echo "== skip"; exit 0
timeout 300 python bench.py --workload c5 --rows 10000000 --steps 20 --warmup 3 --no-cpu-baseline --configs none --no-pmc --f16-slots 3 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c5 shard slots=3', round(d['value']), 'QPS', round(d['ms_per_step'],3), 'ms/step')"
du -sh $O
