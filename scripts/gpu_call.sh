#!/bin/bash
O=gpurun_out/r02k1hf
mkdir -p $O
cd $GRAFT_REPO_ROOT
( time timeout 900 python -m pytest tests/test_two_stage_gpu.py tests/test_random_gpu.py tests/test_full_size_gpu.py::test_ns_full_size_two_stage_equals_the_fp32_scan tests/test_stress_gpu.py -m gpu -x -q -p no:cacheprovider ) > $O/pytest.log 2>&1; grep "passed\|failed\|rror" $O/pytest.log | head -5
timeout 300 python scripts/two_stage_breakdown.py 2>&1 | tail -3
