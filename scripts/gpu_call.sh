#!/bin/bash
O=gpurun_out/r02grow
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 300 python scripts/two_stage_breakdown.py 2>&1 | tail -3
timeout 600 python scripts/bench_two_stage.py > $O/bench_two_stage.log 2>&1; grep -v "^{" $O/bench_two_stage.log | tail -5
( time timeout 900 python -m pytest tests/test_two_stage_gpu.py tests/test_vector_f16_gpu.py tests/test_batcher_gpu.py -m gpu -x -q -p no:cacheprovider ) > $O/pytest.log 2>&1; grep "passed\|failed" $O/pytest.log
