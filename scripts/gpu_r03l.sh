#!/bin/bash
# hybrid tail (one-call hybrid on the range scorer): parity suites + the C4 leg (plain + shadow)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03l
mkdir -p $O
timeout 1500 python -m pytest tests/test_bm25_ranges_gpu.py tests/test_token_score_gpu.py tests/test_two_stage_gpu.py tests/test_allow_resident_gpu.py tests/test_shard_group_gpu.py tests/test_stress_gpu.py -q -m gpu -p no:cacheprovider -x 2>&1 | tail -15 | tee $O/pytest.log
timeout 900 python -m pytest tests/test_full_size_gpu.py -q -m gpu -p no:cacheprovider -x -k "c4" 2>&1 | tail -5 | tee $O/pytest_c4.log
timeout 600 python bench.py --no-pmc --configs c4 2>&1 | tail -1 > $O/bench_c4.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/r03l/bench_c4.json"))
c=d["configs"]["c4"]
print("NS", d["value"], "two-stage", d["two_stage_exact"]["value"])
print("C4", c["value"], c["ms_per_step"], c["full_text_leg"], c["shadow_store"], c["roofline"]["avg_launch_ms"])
PY
ORAMA_BM25_RANGES_HYBRID=0 timeout 600 python bench.py --no-pmc --configs c4 --no-cpu-baseline 2>&1 | tail -1 > $O/bench_c4_k3.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/r03l/bench_c4_k3.json"))
c=d["configs"]["c4"]
print("K3 form: C4", c["value"], c["ms_per_step"], c["full_text_leg"], c["shadow_store"], c["roofline"]["avg_launch_ms"])
PY
