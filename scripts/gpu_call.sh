#!/bin/bash
O=gpurun_out/r02k1hf
mkdir -p $O
cd $GRAFT_REPO_ROOT
( time timeout 900 python -m pytest tests/test_two_stage_gpu.py tests/test_random_gpu.py tests/test_full_size_gpu.py::test_ns_full_size_two_stage_equals_the_fp32_scan tests/test_full_size_gpu.py::test_two_stage_candidates_crowded_into_one_wave tests/test_stress_gpu.py -m gpu -x -q -p no:cacheprovider ) > $O/pytest.log 2>&1; grep "passed\|failed\|rror" $O/pytest.log | head -5; grep -B5 "^E " $O/pytest.log | head -30
for d in 0 2; do ORAMA_K1H_DBG=$d timeout 300 python scripts/two_stage_breakdown.py 2>&1 | tail -3 | head -1; done
