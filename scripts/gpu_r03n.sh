#!/bin/bash
cd $GRAFT_REPO_ROOT
R=$PWD
O=$R/gpurun_out/r03n
mkdir -p $O
timeout 1500 python -m pytest tests/test_bm25_ranges_gpu.py tests/test_token_score_gpu.py tests/test_two_stage_gpu.py tests/test_topn_gpu.py tests/test_vector_gpu.py tests/test_fulltext_gpu.py tests/test_shard_group_gpu.py -q -m gpu -p no:cacheprovider -x 2>&1 | tail -15 | tee $O/pytest.log
export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace -d $O/trace -o h -- python $R/scripts/hybrid_tail_run.py > $O/run.log 2>&1
cd $R
python scripts/hybrid_tail_run.py --report $O/trace 2>&1 | tail -32 | tee $O/hybrid_tail_timeline.log
find $O -name "*.db" -delete
timeout 600 python bench.py --no-pmc --configs c4 2>&1 | tail -1 > $O/bench_c4.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/r03n/bench_c4.json"))
c=d["configs"]["c4"]
print("NS", d["value"], d["roofline"]["topk_select_ms_per_step"], "two-stage", d["two_stage_exact"]["value"])
print("C4", c["value"], c["ms_per_step"], c["full_text_leg"], c["shadow_store"], c["roofline"]["avg_launch_ms"])
PY
