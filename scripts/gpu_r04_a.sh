#!/bin/bash
# Round 4, first GPU call: the sort-free K3r kernel (parity first, then A/B against the merge tree in the comparison
# library), the whole GPU suite (incl. the 8-rank C5 job and the world-8 multi-rank jobs over the loopback transport),
# the driver's bench command, fp16 slots experiment.
set +e
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
O=$R/gpurun_out/r04a
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
echo "== pytest K3r-facing tests first"
timeout 900 python -m pytest tests/test_bm25_ranges_gpu.py tests/test_fulltext_gpu.py tests/test_random_gpu.py tests/test_post_append_gpu.py tests/test_facets_gpu.py tests/test_token_score_gpu.py -q -m gpu -p no:cacheprovider -x 2>&1 | tail -25 | tee $O/pytest_k3r.log
echo "== k3r A/B (comparison library)"
timeout 600 python scripts/k3r_ab.py 2>&1 | tail -30 | tee $O/k3r_ab.log
echo "== pytest -m gpu (all)"
timeout 2400 python -m pytest tests -q -m gpu -p no:cacheprovider --durations=8 2>&1 | tail -30 | tee $O/pytest_gpu.log
echo "== bench (driver command)"
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_all.json 2> $O/bench_all.err; tail -c 600 $O/bench_all.json; echo; tail -5 $O/bench_all.err
echo "== fp16 slots experiment"
for S in 1 2; do
  timeout 300 python bench.py --workload c5 --rows 10000000 --steps 20 --warmup 3 --no-cpu-baseline --configs none --no-pmc --f16-slots $S > $O/c5shard_slots$S.json 2>$O/c5shard_slots$S.err
  python - <<PY
import json
d = json.loads(open("$O/c5shard_slots$S.json").read().strip().splitlines()[-1])
print("c5 shard slots=$S", round(d["value"]), "QPS", round(d["ms_per_step"], 3), "ms/step; scan avg launch", round(d["roofline"]["avg_launch_ms"], 4), "select ms/step", round(d["roofline"]["topk_select_ms_per_step"], 3))
PY
  timeout 300 python bench.py --workload c3 --steps 30 --warmup 3 --no-cpu-baseline --configs none --no-pmc --f16-slots $S > $O/c3_slots$S.json 2>$O/c3_slots$S.err
  python - <<PY
import json
d = json.loads(open("$O/c3_slots$S.json").read().strip().splitlines()[-1])
print("c3 slots=$S", round(d["value"]), "QPS", round(d["ms_per_step"], 3), "ms/step; scan avg launch", round(d["roofline"]["avg_launch_ms"], 4), "select ms/step", round(d["roofline"]["topk_select_ms_per_step"], 3))
PY
done 2>&1 | tee $O/f16_slots.log
du -sh $O
