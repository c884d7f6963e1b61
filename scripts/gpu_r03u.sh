#!/bin/bash
cd $GRAFT_REPO_ROOT
R=$PWD
O=$R/gpurun_out/r03u
mkdir -p $O
timeout 1500 python -m pytest tests/test_bm25_ranges_gpu.py tests/test_token_score_gpu.py tests/test_fulltext_gpu.py tests/test_random_gpu.py tests/test_batcher_gpu.py tests/test_topn_gpu.py tests/test_vector_gpu.py tests/test_vector_f16_gpu.py tests/test_two_stage_gpu.py -q -m gpu -p no:cacheprovider -x 2>&1 | tail -8 | tee $O/pytest.log
python scripts/k3r_chunk_probe.py 2>&1 | tee $O/probe.log
timeout 600 python bench.py --no-pmc --configs c4 --no-cpu-baseline 2>&1 | tail -1 > $O/bench_c4.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/r03u/bench_c4.json"))
c=d["configs"]["c4"]
print("NS", d["value"], "two-stage", d["two_stage_exact"]["value"])
print("C4", c["value"], c["ms_per_step"], c["shadow_store"]["value"])
b=c["bm25_only"]
print("BM25 batch", b["value"], "py", b["through_python_wrapper"]["value"], "single", b["single_query_calls"], "dev us", b["roofline"]["device_us_per_query"], b["roofline"]["device_us_by_kernel"])
PY
